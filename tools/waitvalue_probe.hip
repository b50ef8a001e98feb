// developer probe: can a stream wait for PART of a kernel that runs on another stream?
//   producer (stream A): `nwg` workgroups, each busy for ~`busy_us`; the first `early` of them (in dispatch order) bump a counter
//   consumer (stream B): hipStreamWaitValue32(counter >= early), then a one-wave kernel that stamps the time
// Reports when the counter reached `early`, when the consumer ran, and when the producer ended (all s_memrealtime, 100 MHz),
// for the two kinds of memory the counter can live in, and for a one-wave spinning kernel in place of the stream wait.
//   hipcc --offload-arch=gfx950 -O2 tools/waitvalue_probe.hip -o tools/waitvalue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void producer(unsigned* counter, int early, unsigned long long busy_ticks, unsigned long long* stamp) {
    extern __shared__ double fill[];                      // 64 KB: two workgroups per CU, like the trailing-update kernel
    fill[threadIdx.x] = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < busy_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        if ((int)blockIdx.x < early) {
            __threadfence();
            const unsigned old = atomicAdd(counter, 1u);
            if ((int)old == early - 1) stamp[0] = __builtin_amdgcn_s_memrealtime();   // the counter is complete
        }
        atomicMax(&stamp[2], __builtin_amdgcn_s_memrealtime());                        // the producer's last workgroup
    }
}

__global__ void consumer(unsigned long long* stamp) {
    if (threadIdx.x == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();
}

__global__ void spin_wait(const unsigned* counter, unsigned want) {
    if (threadIdx.x == 0) {
        int guard = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && ++guard < (1 << 22)) __builtin_amdgcn_s_sleep(16);
    }
}

int main() {
    CHECK(hipFuncSetAttribute((const void*)producer, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    int can = -1;
    CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t sa, sb;
    int lo, hi;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    unsigned long long* stamp;
    CHECK(hipMalloc(&stamp, 32));
    unsigned* c_dev;
    unsigned* c_sig;
    CHECK(hipMalloc(&c_dev, 64));
    hipError_t es = hipExtMallocWithFlags((void**)&c_sig, 64, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal) -> %s\n", hipGetErrorString(es));
    const int nwg = 4096, early = 512;
    const unsigned long long busy = 2000;     // 20 us per workgroup: 4096 workgroups on 256 CUs x 2 slots -> 8 waves of the grid, the counter fills in the first
    for (int mode = 0; mode < 3; ++mode) {
        unsigned* counter = (mode == 1 && es == hipSuccess) ? c_sig : c_dev;
        const char* name = mode == 0 ? "stream wait, hipMalloc counter" : mode == 1 ? "stream wait, signal-memory counter" : "spinning one-wave kernel";
        if (mode == 1 && es != hipSuccess) continue;
        if (mode < 2 && !can) continue;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipMemsetAsync(counter, 0, 8, sa));
            CHECK(hipMemsetAsync(stamp, 0, 32, sa));
            CHECK(hipStreamSynchronize(sa));
            // consumer side enqueued first, as the factorisation does with its look-ahead stream
            if (mode < 2) {
                hipError_t e = hipStreamWaitValue32(sb, counter, early, hipStreamWaitValueGte, 0xFFFFFFFFu);
                if (e != hipSuccess) { printf("%s: hipStreamWaitValue32 -> %s\n", name, hipGetErrorString(e)); break; }
            } else {
                hipLaunchKernelGGL(spin_wait, dim3(1), dim3(64), 0, sb, counter, (unsigned)early);
            }
            hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, sb, stamp);
            hipLaunchKernelGGL(producer, dim3(nwg), dim3(256), 65536, sa, counter, early, busy, stamp);
            CHECK(hipStreamSynchronize(sa));
            CHECK(hipStreamSynchronize(sb));
            unsigned long long h[4];
            CHECK(hipMemcpy(h, stamp, 32, hipMemcpyDeviceToHost));
            printf("%-36s counter full -> consumer ran: %7.1f us;  consumer ran %7.1f us before the producer's end\n", name,
                   ((double)h[1] - (double)h[0]) / 100.0, ((double)h[2] - (double)h[1]) / 100.0);
        }
    }
    return 0;
}
