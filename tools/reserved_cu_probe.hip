// developer probe (round 3): would the panel chain run at its stand-alone speed if the trailing update left a set of CUs alone?
// A persistent gemm-like load (MFMA + LDS reads, 64 KB of LDS per workgroup = two per CU, the trailing-update kernel's footprint)
// whose workgroups EXIT at once when they find themselves on a reserved CU (the first `reserve_per_se` CUs of every shader engine
// of every XCD, read from HW_REG_HW_ID / HW_REG_XCC_ID), so that those CUs stay empty while the load runs; the Python side
// (tools/reserved_cu_probe.py) times a small factorisation of the library beside it.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/reserved_cu_probe.hip -o tools/libreserved_probe.so
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define MFMA_V(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

__device__ __forceinline__ unsigned hw_cu(unsigned* xcc) {
    const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID, 16 bits
    *xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15;              // HW_REG_XCC_ID
    return hw;
}

__global__ __launch_bounds__(256) void load_kernel(int iters, int reserve_per_se, const double* in, double* out, unsigned* stats) {
    __shared__ double fill[8192];
    unsigned xcc;
    const unsigned hw = hw_cu(&xcc);
    const unsigned cu = (hw >> 8) & 15;
    if ((int)cu < reserve_per_se) {                      // workgroup-uniform: all waves of a workgroup sit on one CU
        if (threadIdx.x == 0) atomicAdd(&stats[0], 1u);
        return;
    }
    if (threadIdx.x == 0) atomicAdd(&stats[1], 1u);
    for (int i = threadIdx.x; i < 8192; i += 256) fill[i] = 0.0;
    __syncthreads();
    const double2* lds2 = reinterpret_cast<const double2*>(fill) + threadIdx.x;
    double2 acc_l = {0, 0};
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    double a0 = in[threadIdx.x], a1 = in[256 + threadIdx.x], b0 = a1 + 1, b1 = a0 - 1, b2 = a0 + a1, b3 = a0 - a1;
    for (int it = 0; it < iters; ++it) {
        if ((it & 3) == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { const double2 v = lds2[256 * (q & 7)]; acc_l.x += v.x; acc_l.y += v.y; }
        }
        MFMA_V(c0, a0, b0); MFMA_V(c1, a0, b1); MFMA_V(c2, a0, b2); MFMA_V(c3, a0, b3);
        MFMA_V(c4, a1, b0); MFMA_V(c5, a1, b1); MFMA_V(c6, a1, b2); MFMA_V(c7, a1, b3);
    }
    double4_t s4 = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s4[0] + s4[1] + s4[2] + s4[3] + acc_l.x + acc_l.y == 1.2345) out[0] = s4[0];
}

static hipStream_t g_stream = nullptr;
static double *g_in = nullptr, *g_out = nullptr;
static unsigned* g_stats = nullptr;

extern "C" int probe_load_start(int iters, int reserve_per_se, int workgroups) {
    if (!g_stream) {
        if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
        if (hipMalloc(&g_in, 512 * 8) != hipSuccess || hipMalloc(&g_out, 64) != hipSuccess || hipMalloc(&g_stats, 16) != hipSuccess) return -2;
        double h[512];
        for (int i = 0; i < 512; ++i) h[i] = 0.5;
        if (hipMemcpy(g_in, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return -3;
    }
    if (hipMemsetAsync(g_stats, 0, 16, g_stream) != hipSuccess) return -4;
    hipLaunchKernelGGL(load_kernel, dim3(workgroups), dim3(256), 0, g_stream, iters, reserve_per_se, g_in, g_out, g_stats);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// 1 while the load is still running
extern "C" int probe_load_running() { return hipStreamQuery(g_stream) == hipErrorNotReady ? 1 : 0; }

extern "C" int probe_load_wait(unsigned* exited, unsigned* stayed) {
    if (hipStreamSynchronize(g_stream) != hipSuccess) return -1;
    unsigned h[4] = {0, 0, 0, 0};
    if (hipMemcpy(h, g_stats, 16, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    *exited = h[0];
    *stayed = h[1];
    return 0;
}

// a chain of n dependent trivial launches (one workgroup of 64 threads each) on a stream of its own: milliseconds for all of them.
// Beside the load this isolates what a dependent launch costs when another grid is resident - no arithmetic, no LDS to speak of.
__global__ void tiny_kernel(unsigned* p) { if (threadIdx.x == 0) p[2] += 1u; }
__global__ void chain_fma_kernel(int n, const double* in, double* out) {        // one wave, n dependent fp64 fmas: the clock it runs at
    double a = in[threadIdx.x] * 1e-3, b = 1.0000001;
    for (int i = 0; i < n; ++i) a = fma(a, b, 1e-9);
    if (a == 1.2345) out[0] = a;
}
static hipStream_t g_chain = nullptr;
static hipEvent_t g_e0, g_e1;
extern "C" double probe_tiny_chain(int n, int fma_len) {
    if (!g_chain) {
        int lo, hi;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&g_chain, hipStreamNonBlocking, hi) != hipSuccess) return -1.0;
        hipEventCreate(&g_e0);
        hipEventCreate(&g_e1);
    }
    hipEventRecord(g_e0, g_chain);
    for (int i = 0; i < n; ++i) {
        if (fma_len > 0) hipLaunchKernelGGL(chain_fma_kernel, dim3(1), dim3(64), 0, g_chain, fma_len, g_in, g_out);
        else hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, g_chain, g_stats);
    }
    hipEventRecord(g_e1, g_chain);
    if (hipEventSynchronize(g_e1) != hipSuccess) return -2.0;
    float ms = 0.f;
    hipEventElapsedTime(&ms, g_e0, g_e1);
    return (double)ms;
}
