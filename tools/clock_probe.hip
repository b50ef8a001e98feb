// developer probe: clock held during short single-wave kernels launched back to back with idle gaps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
__global__ void k(int n, double* out, unsigned long long* clk) {
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) a = fma(a, b, 1e-9);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    if (a == 1.2345) out[0] = a;
}
int main() {
    double* out; unsigned long long* clk; hipMalloc(&out, 8); hipMalloc(&clk, 16);
    unsigned long long h[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int gap_us : {0, 100, 1000, 20000}) {
        for (int rep = 0; rep < 4; ++rep) {
            if (gap_us) usleep(gap_us);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 4000, out, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("gap %6d us: 4000 dependent fp64 FMAs: %llu cycles, %.1f us in-kernel (clock %.2f GHz, %.1f cycles/FMA), event time %.1f us\n",
                   gap_us, h[0], h[1] / 100.0, (double)h[0] / h[1] * 0.1, h[0] / 4000.0, ms * 1e3);
        }
    }
    return 0;
}
