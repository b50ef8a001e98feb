// developer probe: operand layout of v_mfma_f64_4x4x4_4b_f64 (64 lanes = 4 blocks of a 4x4x4 product, one double per lane for A, B, C/D).
// For every pair (a, b) of lanes inside block 0: A = unit at lane a, B = unit at lane b -> which lane of D receives the 1?
// Prints the (i, k) of an A lane, the (k, j) of a B lane, the (i, j) of a D lane, and checks the identity the panel kernels would
// use: with A = coef x I (coef on the A lanes of the diagonal), D = coef * B + C lane by lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int* out, double* chk) {
    const int l = threadIdx.x;
    for (int a = 0; a < 16; ++a)
        for (int b = 0; b < 16; ++b) {
            const double A = (l == a) ? 1.0 : 0.0, B = (l == b) ? 1.0 : 0.0;
            const double D = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, 0.0, 0, 0, 0);
            if (D != 0.0) out[a * 16 + b] = l;
        }
    // coef x I on lanes 0, 5, 10, 15 of every block (coefficient differs per block), B = lane-specific data, C = lane-specific
    const int blk = l >> 4, w = l & 15;
    const double coef = 2.0 + blk;
    const double A = (w == 0 || w == 5 || w == 10 || w == 15) ? coef : 0.0;
    const double B = 100.0 + l, C = 0.25 * l;
    chk[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, C, 0, 0, 0) - (coef * B + C);
}
int main() {
    int* out; double* chk;
    hipMalloc(&out, 256 * 4); hipMemset(out, 0xff, 256 * 4); hipMalloc(&chk, 64 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, chk);
    int h[256]; double c[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(c, chk, sizeof(c), hipMemcpyDeviceToHost);
    printf("D lane for (A unit lane a, B unit lane b), -1 = no contribution:\n      ");
    for (int b = 0; b < 16; ++b) printf("b%-3d", b);
    printf("\n");
    for (int a = 0; a < 16; ++a) { printf("a%-3d: ", a); for (int b = 0; b < 16; ++b) printf("%-4d", h[a * 16 + b]); printf("\n"); }
    double worst = 0; for (int l = 0; l < 64; ++l) if (fabs(c[l]) > worst) worst = fabs(c[l]);
    printf("A = coef x I (lanes 0, 5, 10, 15 of each block): max |D - (coef * B + C)| over the 64 lanes = %g\n", worst);
    return 0;
}
