// Device-side evaluation of one covariance entry k(x1, x2) for a flattened sum of kernel terms.
// Restates mgpi/kernels.py:128-131 (WhiteNoise), :180-185 (SquaredExponential, no factor 1/2) and
// :287-293 (CombinedKernel: ans = 0.0; ans += term, in summand order).
#pragma once
#include "mgpi_internal.h"

// Matern shape function  m(x) = 2^(1-o)/Gamma(o) * x^o * K_o(x),  x > 0, real order o = nl + mu (|mu| <= 1/2).
// K_mu and K_{mu+1} come from Temme's series for x < 2 and Steed's continued fraction (CF2) above (Temme 1975; Press et
// al. 6.7 describe the same scheme); the climb to order o uses the recurrence K_{v+1} = K_{v-1} + (2v/x) K_v - stable
// upwards for K - written for T_v = x^v K_v:  T_{v+1} = x^2 T_{v-1} + 2v T_v.  T_v is positive, decreasing in x and
// bounded by 2^(v-1) Gamma(v), so neither x^o (huge for far pairs) nor K_o (huge for near pairs) is ever formed on its
// own.  The order-only constants (gam1, gam2, 1/Gamma(1 +- mu), norm = 2^(1-o)/Gamma(o)) come precomputed from the host.
// Takes the role of  (2**(1-o)/gamma(o)) * diff**o * scipy.special.kv(o, diff)  in mgpi/kernels.py:162.
__device__ __forceinline__ double mgpi_matern_shape(double x, int nl, double mu, double norm, double gam1, double gam2,
                                                    double gampl, double gammi) {
    if (nl > 0 && x < 1e-100) return 1.0;                    // 1 - m(x) = O(x^min(2o,2)): below one ulp here; K_{mu+1} would overflow
    const double eps = 1e-16, mu2 = mu * mu, xi = 1.0 / x;
    double kmu, k1;
    if (x < 2.0) {
        // Temme: f_0 = (mu pi / sin mu pi) [ cosh(sg) Gamma_1 + (sinh sg / sg) ln(2/x) Gamma_2 ],  sg = mu ln(2/x)
        const double l2x = -log(0.5 * x);                    // ln(2/x)
        const double sg = mu * l2x;
        const double shc = (fabs(sg) < eps) ? 1.0 : sinh(sg) / sg;
        const double pimu = 3.141592653589793 * mu;
        const double fact = (fabs(pimu) < eps) ? 1.0 : pimu / sin(pimu);
        double ff = fact * (gam1 * cosh(sg) + gam2 * shc * l2x);
        double sum = ff;
        const double ee = exp(sg);                           // (2/x)^mu
        double p = 0.5 * ee / gampl;                         // p_0 = (1/2) (x/2)^-mu Gamma(1+mu)
        double q = 0.5 / (ee * gammi);                       // q_0 = (1/2) (x/2)^+mu Gamma(1-mu)
        double c = 1.0;
        const double d = 0.25 * x * x;
        double sum1 = p;
        for (int i = 1; i <= 500; ++i) {
            ff = (i * ff + p + q) / (i * (double)i - mu2);
            c *= d / i;
            p /= (i - mu);
            q /= (i + mu);
            const double del = c * ff;
            sum += del;
            sum1 += c * (p - i * ff);
            if (fabs(del) < fabs(sum) * eps) break;
        }
        kmu = sum;
        k1 = sum1 * 2.0 * xi;
    } else {
        double b = 2.0 * (1.0 + x), d = 1.0 / b, h = d, delh = d;
        double q1 = 0.0, q2 = 1.0;
        const double a1 = 0.25 - mu2;
        double q = a1, c = a1, a = -a1;
        double s = 1.0 + q * delh;
        for (int i = 2; i <= 500; ++i) {
            a -= 2 * (i - 1);
            c = -a * c / i;
            const double qnew = (q1 - b * q2) / a;
            q1 = q2;
            q2 = qnew;
            q += c * qnew;
            b += 2.0;
            d = 1.0 / (b + a * d);
            delh = (b * d - 1.0) * delh;
            h += delh;
            const double dels = q * delh;
            s += dels;
            if (fabs(dels) < fabs(s) * eps) break;
        }
        h = a1 * h;
        kmu = sqrt(3.141592653589793 / (2.0 * x)) * exp(-x) / s;
        k1 = kmu * (mu + x + 0.5 - h) * xi;
    }
    const double xm = pow(x, mu);
    double t0 = xm * kmu;                                    // T_mu
    if (nl > 0) {
        double t1 = xm * x * k1;                             // T_{mu+1}
        const double x2 = x * x;
        for (int i = 1; i < nl; ++i) {
            const double tn = x2 * t0 + 2.0 * (mu + i) * t1; // T_{mu+i+1}
            t0 = t1;
            t1 = tn;
        }
        t0 = t1;
    }
    return norm * t0;
}

// a / b correctly rounded from rb = RN(1/b): q = RN(a rb), the residual a - q b is exact in an fma, one correction step
// (Markstein).  Holds for normal finite b (mgpi_terms_dev::exact_div); the reference divides (kernels.py:185), and
// an error of one ulp in r = sum (dx/l)^2 becomes r ulps in exp(-r) - the reason round 1 could only promise 5e-13 on K.
__device__ __forceinline__ double mgpi_div_rn(double a, double b, double rb) {
    const double q = __dmul_rn(a, rb);
    return fma(fma(-q, b, a), rb, q);
}

// exp(-r) for r >= 0 (also NaN -> NaN, +inf -> 0), inlined: k = rint(-r / ln 2), t = -r - k ln 2 in two fma steps
// (ln2_hi carries 32 trailing zero bits: k ln2_hi is exact for |k| < 2^11), a degree-13 Taylor polynomial on
// |t| <= ln2 / 2 (remainder t^14 / 14! < 5e-18), scaling by 2^k in one v_ldexp_f64 (gradual underflow like libm).
// 21 instructions against ~45 for the library call, whose argument checks and two-sided range handling this
// one-sided form does not need.  Measured against numpy's exp over r in [0, 745]: max 1 ulp (tests/test_gpu_parity.py).
__device__ __forceinline__ double mgpi_exp_neg(double r) {
    const double x = -r;
    const double kf = __builtin_rint(x * 1.4426950408889634);
    double t = fma(-kf, 6.93147180369123816490e-01, x);
    t = fma(-kf, 1.90821492927058770002e-10, t);
    double p = 1.6059043836821613e-10;                         // 1/13!
    p = fma(p, t, 2.08767569878681e-09);                       // 1/12!
    p = fma(p, t, 2.505210838544172e-08);                      // 1/11!
    p = fma(p, t, 2.755731922398589e-07);                      // 1/10!
    p = fma(p, t, 2.7557319223985893e-06);                     // 1/9!
    p = fma(p, t, 2.48015873015873e-05);                       // 1/8!
    p = fma(p, t, 1.984126984126984e-04);                      // 1/7!
    p = fma(p, t, 1.388888888888889e-03);                      // 1/6!
    p = fma(p, t, 8.333333333333333e-03);                      // 1/5!
    p = fma(p, t, 4.1666666666666664e-02);                     // 1/4!
    p = fma(p, t, 1.6666666666666666e-01);                     // 1/3!
    p = fma(p, t, 0.5);
    p = fma(p, t, 1.0);
    p = fma(p, t, 1.0);
    const double v = ldexp(p, (int)kf);
    return (r > 746.0) ? 0.0 : v;                              // below the smallest subnormal (and r = +inf, where t is NaN)
}

// the two commonest kernels without the term loop: MIX 1 = one SquaredExponential, MIX 2 = SquaredExponential +
// WhiteNoise (kernel + nugget, kernels.py:54-59).  Same operations in the same order as the general loop below.
template <int D, int MIX>
__device__ __forceinline__ double mgpi_cov_entry_fast(const mgpi_terms_dev& t, const double* __restrict__ a,
                                                      const double* __restrict__ b) {
    const int nd = (D > 0) ? D : t.d;                              // D == 0: run-time dimensionality, still no term loop
    double r = 0.0;
    bool same = true;
#pragma unroll
    for (int k = 0; k < ((D > 0) ? D : MGPI_MAX_D); ++k) {
        if (k < nd) {
            const double df = a[k] - b[k];
            same = same && (a[k] == b[k]);
            r = __dadd_rn(r, mgpi_div_rn(__dmul_rn(df, df), t.l2[0][k], t.inv_l2[0][k]));
        }
    }
    double val = __dmul_rn(t.sig2[0], mgpi_exp_neg(r));
    if (MIX == 2) val = __dadd_rn(val, same ? t.sig2[1] : 0.0);
    return val;
}

// D > 0: compile-time dimensionality (registers, fully unrolled); D == 0: runtime t.d (<= MGPI_MAX_D)
// summands [lo, hi) of the kernel: the NNGP path needs kernel-only and nugget-only sums next to the full one
template <int D>
__device__ __forceinline__ double mgpi_cov_entry_range(const mgpi_terms_dev& t, const double* __restrict__ a,
                                                       const double* __restrict__ b, int lo, int hi) {
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
    for (int s = lo; s < hi; ++s) {
        if (t.type[s] == MGPI_TERM_SE) {
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < ((D > 0) ? D : MGPI_MAX_D); ++k)
                if (k < nd)   // sum_d (x1-x2)**2/length_d**2, left to right; the quotient correctly rounded like the reference's division
                    r = __dadd_rn(r, t.exact_div ? mgpi_div_rn(diff2[k], t.l2[s][k], t.inv_l2[s][k]) : diff2[k] / t.l2[s][k]);
            val = __dadd_rn(val, __dmul_rn(t.sig2[s], mgpi_exp_neg(r)));   // sigma**2 * exp(-r), rounded product then add
        } else if (t.type[s] == MGPI_TERM_MATERN) {
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < ((D > 0) ? D : MGPI_MAX_D); ++k)
                if (k < nd) r = __dadd_rn(r, t.exact_div ? mgpi_div_rn(diff2[k], t.l2[s][k], t.inv_l2[s][k]) : diff2[k] / t.l2[s][k]);
            const double diff = sqrt(t.mat_two_o[s]) * sqrt(r);    // (2*o)**0.5 * sum(...)**0.5   (kernels.py:161)
            double m = 1.0;                                        // r -> 0 limit of the bracket below (the reference: NaN)
            if (t.mat_half_p[s] >= 0) {
                // half-integer order: exp(-x) times a polynomial of degree p (Horner), no special function needed
                double poly = 0.0;
                for (int i = t.mat_half_p[s]; i >= 0; --i) poly = fma(poly, diff, t.mat_poly[s][i]);
                m = exp(-diff) * poly;                             // = 1 at diff == 0
            } else if (diff > 0.0)
                m = mgpi_matern_shape(diff, t.mat_nl[s], t.mat_mu[s], t.mat_norm[s], t.mat_gam1[s], t.mat_gam2[s],
                                      t.mat_gampl[s], t.mat_gammi[s]);
            val = __dadd_rn(val, __dmul_rn(t.sig2[s], m));
        } else {
            val = __dadd_rn(val, same ? t.sig2[s] : 0.0);        // sigma**2 * all(x1 == x2)
        }
    }
    return val;
}

// MIX is a compile-time copy of t.fast_mix (the launchers switch on it)
template <int D, int MIX = 0>
__device__ __forceinline__ double mgpi_cov_entry(const mgpi_terms_dev& t, const double* __restrict__ a,
                                                 const double* __restrict__ b) {
    if constexpr (MIX > 0) return mgpi_cov_entry_fast<D, MIX>(t, a, b);
    else return mgpi_cov_entry_range<D>(t, a, b, 0, t.nterms);
}
