// Device-side evaluation of one covariance entry k(x1, x2) for a flattened sum of kernel terms.
// Restates mgpi/kernels.py:128-131 (WhiteNoise), :180-185 (SquaredExponential, no factor 1/2) and
// :287-293 (CombinedKernel: ans = 0.0; ans += term, in summand order).
#pragma once
#include "mgpi_internal.h"

// D > 0: compile-time dimensionality (registers, fully unrolled); D == 0: runtime t.d (<= MGPI_MAX_D)
template <int D>
__device__ __forceinline__ double mgpi_cov_entry(const mgpi_terms_dev& t, const double* __restrict__ a,
                                                 const double* __restrict__ b) {
    const int nd = (D > 0) ? D : t.d;
    // coordinate differences are shared by all terms; (x1 == x2) <=> (x1 - x2 == 0) for finite fp64
    double diff2[(D > 0) ? D : MGPI_MAX_D];
    bool same = true;
#pragma unroll
    for (int k = 0; k < ((D > 0) ? D : MGPI_MAX_D); ++k) {
        if (k < nd) {
            const double df = a[k] - b[k];
            same = same && (a[k] == b[k]);
            diff2[k] = __dmul_rn(df, df);                        // (x1-x2)**2
        }
    }
    double val = 0.0;
    for (int s = 0; s < t.nterms; ++s) {
        if (t.type[s] == MGPI_TERM_SE) {
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < ((D > 0) ? D : MGPI_MAX_D); ++k)
                if (k < nd) r = __dadd_rn(r, __dmul_rn(diff2[k], t.inv_l2[s][k]));   // sum_d (x1-x2)**2/length_d**2, left to right, no fma contraction
            val = __dadd_rn(val, __dmul_rn(t.sig2[s], exp(-r)));   // sigma**2 * exp(-r), rounded product then add
        } else {
            val = __dadd_rn(val, same ? t.sig2[s] : 0.0);        // sigma**2 * all(x1 == x2)
        }
    }
    return val;
}
