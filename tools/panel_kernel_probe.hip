// developer probe: correctness against a host Cholesky / substitution and shader-clock timing of the phases of the
// accumulator-layout panel primitives (one workgroup, stand-alone).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I muses-gaussian-process-interpolator_amd/csrc tools/panel_kernel_probe.hip -o tools/panel_kernel_probe
#define PANEL_DEBUG 1
#define PANEL_STAMPS 1
__device__ long long panel_stamps[8];
__device__ double panel_debug[16];
__device__ double panel_debug2[256 * 8];
#include "panel_device.h"
#include <vector>
#include <math.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void probe_kernel(double* __restrict__ A, int64_t lda, const double* __restrict__ diag0,
                                                    double* __restrict__ X, double* __restrict__ S, long long* __restrict__ stamps, int mode, double* __restrict__ dump) {
    __shared__ panel_lds s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    long long t0 = clock64();
    panel_double4_t accD[4], accT[4];
    panel_load(accD, A, lda, w, g, c, true);
    panel_load(accT, X, IB, w, g, c, false);
    if (tid < IB) s.dtol[tid] = 2.2e-14 * fabs(diag0[tid]);
    __syncthreads();
    long long t1 = clock64();
    int bad = 0;
    if (mode & 1) potf2_acc_rounds<0>::run(accD, s, w, g, c, bad, S);
    __syncthreads();
    long long t2 = clock64();
    if (mode & 2) { panel_trsm_images(s, tid); trsm_acc_mx<0>::run(accT, s, w, g, c, X + (int64_t)(16 * w + c) * IB); }
    long long t3 = clock64();
    if (tid == 0) { stamps[0] = t1 - t0; stamps[1] = t2 - t1; stamps[2] = t3 - t2; stamps[3] = bad; }
    if (lane == 5) stamps[4 + w] = bad;
    __syncthreads();
    for (int i = tid; i < IB * IB; i += 256) dump[i] = s.Lt[i / IB][i % IB];
    if (tid < IB) { dump[IB * IB + tid] = s.invd[tid]; dump[IB * IB + IB + tid] = s.dtol[tid]; }
}

__global__ __launch_bounds__(256) void one_round_kernel(double* __restrict__ A, int64_t lda, const double* __restrict__ diag0,
                                                        const double* __restrict__ X, double* __restrict__ out, int fence) {
    __shared__ panel_lds s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    panel_double4_t accD[4], acc[4];
    panel_load(accD, A, lda, w, g, c, true);
    panel_load(acc, X, IB, w, g, c, false);
    if (tid < IB) s.dtol[tid] = 2.2e-14 * fabs(diag0[tid]);
    __syncthreads();
    int bad = 0;
    potf2_acc_rounds<0>::run(accD, s, w, g, c, bad, nullptr);
    __syncthreads();
    constexpr int K = 0, QK = 0, KL = 0;
    if (c >= KL && c < KL + 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s.zraw[w][g + 4 * r][c - KL] = acc[QK][r];
    }
    if (fence) __syncthreads();
    __asm__ volatile("" ::: "memory");
    const panel_double4_t a = panel_row4(s.zraw[w][c]);
    const double l10 = s.Lt[K][K + 1], l20 = s.Lt[K][K + 2], l30 = s.Lt[K][K + 3];
    const double l21 = s.Lt[K + 1][K + 2], l31 = s.Lt[K + 1][K + 3], l32 = s.Lt[K + 2][K + 3];
    const double y0 = a[0] * s.invd[K];
    const double y1 = fma(-y0, l10, a[1]) * s.invd[K + 1];
    const double y2 = fma(-y1, l21, fma(-y0, l20, a[2])) * s.invd[K + 2];
    const double y3 = fma(-y2, l32, fma(-y1, l31, fma(-y0, l30, a[3]))) * s.invd[K + 3];
    const double yg = (g == 0) ? y0 : (g == 1) ? y1 : (g == 2) ? y2 : y3;
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = PANEL_MFMA(-yg, s.Lt[K + g][16 * n + c], acc[n]);
    panel_store(acc, out, IB, w, g, c);
    out[IB * IB + (16 * w + c) * 4 + g] = yg;
}

int main() {
    const int n = 64;
    std::vector<double> K(n * n), X(n * n), d(n), L(n * n, 0.0), Xref(n * n), gotS(n * n), gotX(n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) K[i * n + j] = exp(-0.05 * (i - j) * (i - j)) + (i == j ? 0.1 : 0.0) + 0.05 * sin(1.0 + i) * sin(1.0 + j);
    for (int i = 0; i < n; ++i) d[i] = K[i * n + i];
    for (int i = 0; i < n * n; ++i) X[i] = sin(0.1 * i) + 0.3;
    // host reference
    for (int j = 0; j < n; ++j) {
        double sdiag = K[j * n + j];
        for (int k = 0; k < j; ++k) sdiag -= L[j * n + k] * L[j * n + k];
        L[j * n + j] = sqrt(sdiag);
        for (int i = j + 1; i < n; ++i) {
            double v = K[i * n + j];
            for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = v / L[j * n + j];
        }
    }
    for (int r = 0; r < n; ++r)
        for (int k = 0; k < n; ++k) {
            double v = X[r * n + k];
            for (int m = 0; m < k; ++m) v -= Xref[r * n + m] * L[k * n + m];
            Xref[r * n + k] = v / L[k * n + k];
        }
    double *dK, *dX, *dd, *dS;
    long long* ds;
    double* ddump; CK(hipMalloc(&ddump, sizeof(double) * (n * n + 2 * n)));
    std::vector<double> dump(n * n + 2 * n);
    CK(hipMalloc(&dK, sizeof(double) * n * n)); CK(hipMalloc(&dX, sizeof(double) * n * n)); CK(hipMalloc(&dS, sizeof(double) * n * n));
    CK(hipMalloc(&dd, sizeof(double) * n)); CK(hipMalloc(&ds, 128));
    {
        double* dout; CK(hipMalloc(&dout, sizeof(double) * (n * n + 4 * n)));
        std::vector<double> o(n * n + 4 * n);
        for (int fence = 0; fence < 2; ++fence) {
            CK(hipMemcpy(dK, K.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
            CK(hipMemcpy(dX, X.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
            CK(hipMemcpy(dd, d.data(), sizeof(double) * n, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(one_round_kernel, dim3(1), dim3(256), 0, 0, dK, (int64_t)n, dd, dX, dout, fence);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(o.data(), dout, sizeof(double) * (n * n + 4 * n), hipMemcpyDeviceToHost));
            double ey = 0, eu = 0; int wi = -1, wj = -1;
            for (int i = 0; i < n; ++i) {
                for (int t = 0; t < 4; ++t) ey = fmax(ey, fabs(o[n * n + 4 * i + t] - Xref[i * n + t]));
                for (int j = 4; j < n; ++j) {
                    double want = X[i * n + j];
                    for (int t = 0; t < 4; ++t) want -= Xref[i * n + t] * L[j * n + t];
                    const double e = fabs(o[i * n + j] - want);
                    if (e > eu) { eu = e; wi = i; wj = j; }
                }
            }
            printf("one round (fence %d): y err %.2e, updated-tile err %.2e at (%d, %d)\n", fence, ey, eu, wi, wj);
        }
    }
    for (int mode = 1; mode < 4; mode += 2)
        for (int rep = 0; rep < 2; ++rep) {
            std::vector<double> Kup(K);
            for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) Kup[i * n + j] = NAN;      // the strict upper triangle is never used
            CK(hipMemcpy(dK, Kup.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
            CK(hipMemcpy(dX, X.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
            CK(hipMemcpy(dd, d.data(), sizeof(double) * n, hipMemcpyHostToDevice));
            CK(hipMemset(dS, 0xff, sizeof(double) * n * n));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 0, 0, dK, (int64_t)n, dd, dX, dS, ds, mode, ddump);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            long long st[8];
            CK(hipMemcpy(st, ds, 64, hipMemcpyDeviceToHost));
            CK(hipMemcpy(gotS.data(), dS, sizeof(double) * n * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(gotX.data(), dX, sizeof(double) * n * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(dump.data(), ddump, sizeof(double) * (n * n + 2 * n), hipMemcpyDeviceToHost));
            double eLt = 0, einv = 0;
            if (mode & 1) for (int k = 0; k < n; ++k) { einv = fmax(einv, fabs(dump[n * n + k] - 1.0 / L[k * n + k]));
                for (int m = k; m < n; ++m) eLt = fmax(eLt, fabs(dump[k * n + m] - L[m * n + k])); }
            printf("   per-wave bad %lld %lld %lld %lld | Lt err %.2e invd err %.2e dtol[42] %.3e\n", st[4], st[5], st[6], st[7], eLt, einv, dump[n * n + n + 42]);
            long long ps[8]; CK(hipMemcpyFromSymbol(ps, HIP_SYMBOL(panel_stamps), sizeof(ps)));
            printf("   round 32 (wave 3): write+barrier %lld, reads+pivot chain %lld, solves+mfma+stores %lld cycles (own-row solve %lld, B solves + mfma %lld, rsq + stores %lld)\n", ps[0], ps[1], ps[2], ps[3], ps[4], ps[5]);
            double dbg[16]; CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(panel_debug), sizeof(dbg)));
            printf("   dbg: e = %.4e %.4e %.4e %.4e | tol %.3e %.3e %.3e %.3e | ok %g %g %g %g bad %g\n", dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7], dbg[8], dbg[9], dbg[10], dbg[11], dbg[12]);
            double eL = 0, eX = 0, upper = 0;
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
                if (mode & 1) { eL = fmax(eL, fabs(gotS[i * n + j] - L[i * n + j])); if (j > i) upper = fmax(upper, fabs(gotS[i * n + j])); }
                if (mode == 3) eX = fmax(eX, fabs(gotX[i * n + j] - Xref[i * n + j]));
            }
            if (mode == 3 && rep == 0) {
                std::vector<double> d2(256 * 8); CK(hipMemcpyFromSymbol(d2.data(), HIP_SYMBOL(panel_debug2), sizeof(double) * 256 * 8));
                for (int t = 0; t < 256; t += 37) {
                    const int lane = t & 63, w = t >> 6, g = lane >> 4, c = lane & 15, row = 16 * w + c;
                    double want[4];
                    for (int q = 0; q < 4; ++q) { want[q] = X[row * n + 4 + q]; for (int u = 0; u < 4; ++u) want[q] -= Xref[row * n + u] * L[(4 + q) * n + u]; }
                    printf("   tid %3d (w %d g %d c %2d row %2d): a = %.5f %.5f %.5f %.5f want %.5f %.5f %.5f %.5f | yg %.5f ref %.5f | invd %.4f (%.4f) l10 %.4f (%.4f)\n", t, w, g, c, row,
                           d2[t * 8], d2[t * 8 + 1], d2[t * 8 + 2], d2[t * 8 + 3], want[0], want[1], want[2], want[3], d2[t * 8 + 4], Xref[row * n + 4 + g], d2[t * 8 + 5], 1 / L[4 * n + 4], d2[t * 8 + 6], L[5 * n + 4]);
                }
                printf("   X err per 4-column block:");
                for (int b = 0; b < 16; ++b) { double e = 0; for (int i = 0; i < n; ++i) for (int j = 4 * b; j < 4 * b + 4; ++j) e = fmax(e, fabs(gotX[i * n + j] - Xref[i * n + j])); printf(" %.1e", e); }
                printf("\n   X err per 16-row block (cols 4..7):");
                for (int b = 0; b < 4; ++b) { double e = 0; for (int i = 16 * b; i < 16 * b + 16; ++i) for (int j = 4; j < 8; ++j) e = fmax(e, fabs(gotX[i * n + j] - Xref[i * n + j])); printf(" %.1e", e); }
                printf("\n   row 0: got %.6f %.6f %.6f %.6f %.6f %.6f ref %.6f %.6f %.6f %.6f %.6f %.6f\n", gotX[0], gotX[1], gotX[2], gotX[3], gotX[4], gotX[5], Xref[0], Xref[1], Xref[2], Xref[3], Xref[4], Xref[5]);
            }
            printf("mode %d rep %d: kernel %.1f us | cycles: load %lld potf2 %lld trsm %lld | bad %lld | max|L-Lref| %.2e upper %.1e max|X-Xref| %.2e\n",
                   mode, rep, ms * 1e3, st[0], st[1], st[2], st[3], eL, upper, eX);
        }
    return 0;
}
