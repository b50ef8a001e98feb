// developer probe: fp64 MFMA issue rate and the clock the chip holds under load (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(int iters, const double* in, double* out, unsigned long long* clk) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    const double a = in[threadIdx.x], b = in[256 + threadIdx.x];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    if (s == 1.2345) out[0] = s;
}

template <int NACC>
void run(const char* name, int blocks, int iters, const double* din, double* dout, unsigned long long* dclk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, 100, din, dout, dclk);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, iters, din, dout, dclk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), dclk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double cyc = 0, real = 0;
        for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
        double ghz = cyc / real * 0.1;
        double flops = (double)blocks * 4 * iters * NACC * 2048.0;
        double cyc_per_mfma = (cyc / blocks) / ((double)iters * NACC);
        printf("%-28s blocks=%d: %.2f ms  %.1f TFLOP/s  clock %.2f GHz  %.1f cycles/MFMA/wave\n", name, blocks, ms, flops / ms / 1e9, ghz, cyc_per_mfma);
    }
}

int main() {
    double *din, *dout; unsigned long long* dclk;
    hipMalloc(&din, 512 * 8); hipMalloc(&dout, 8); hipMalloc(&dclk, 16 * 4096);
    std::vector<double> h(512);
    srand(1);
    for (auto& v : h) v = 2.0 * rand() / RAND_MAX - 1.0;
    hipMemcpy(din, h.data(), 512 * 8, hipMemcpyHostToDevice);
    run<4>("random, 1 wave/SIMD, 4 acc", 256, 20000, din, dout, dclk);
    run<4>("random, 2 waves/SIMD, 4 acc", 512, 20000, din, dout, dclk);
    run<4>("random, 3 waves/SIMD, 4 acc", 768, 20000, din, dout, dclk);
    run<4>("random, 4 waves/SIMD, 4 acc", 1024, 20000, din, dout, dclk);
    run<4>("random, 8 waves/SIMD, 4 acc", 2048, 10000, din, dout, dclk);
    run<2>("random, 8 waves/SIMD, 2 acc", 2048, 20000, din, dout, dclk);
    return 0;
}
