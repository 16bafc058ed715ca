// model.hpp - device-side prior densities, bounds check and built-in likelihood families.
//
// Replaces, per particle and per proposal, the reference's
//   update!(parameters, para_new)  (bounds => ParamBoundsError)      src/mutation.jl:93
//   prior(parameters)              (Σ logpdf over free parameters)   src/mutation.jl:95
//   loglikelihood(parameters, data), old_loglikelihood(.., old_data) src/mutation.jl:96,106
// A parameter vector is read through an accessor `th(k)` so the caller can keep it in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "devstate.hpp"

namespace smcmi {

constexpr double LOG2PI = 1.8378770664093454835606594728112;
#define SMCMI_NEG_INF (-__builtin_huge_val())

// host: per-parameter constant so the device never evaluates log() of a constant
inline double prior_const_host(int fam, double a, double b) {
    switch (fam) {
    case SMCMI_PRIOR_NORMAL: return log(b);
    case SMCMI_PRIOR_UNIFORM: return -log(b - a);
    case SMCMI_PRIOR_GAMMA: return -lgamma(a) - a * log(b);
    case SMCMI_PRIOR_BETA: return -(lgamma(a) + lgamma(b) - lgamma(a + b));
    case SMCMI_PRIOR_INVGAMMA: return a * log(b) - lgamma(a);
    case SMCMI_PRIOR_ROOTINVGAMMA: return log(2.0) - lgamma(a / 2.0) + (a / 2.0) * log(a * b * b / 2.0);
    default: return NAN;
    }
}

__device__ inline double prior_logpdf(int fam, double a, double b, double k, double x) {
    switch (fam) {
    case SMCMI_PRIOR_NORMAL: { const double z = (x - a) / b; return -(z * z + LOG2PI) / 2.0 - k; }
    case SMCMI_PRIOR_UNIFORM: return (a <= x && x <= b) ? k : SMCMI_NEG_INF;
    case SMCMI_PRIOR_GAMMA: return x < 0 ? SMCMI_NEG_INF : k + (a - 1.0) * log(x) - x / b;
    case SMCMI_PRIOR_BETA: return (x < 0 || x > 1) ? SMCMI_NEG_INF : (a - 1.0) * log(x) + (b - 1.0) * log1p(-x) + k;
    case SMCMI_PRIOR_INVGAMMA: return x <= 0 ? SMCMI_NEG_INF : k - (a + 1.0) * log(x) - b / x;
    case SMCMI_PRIOR_ROOTINVGAMMA:
        return x <= 0 ? SMCMI_NEG_INF : k - ((a + 1.0) / 2.0) * log(x * x) - a * b * b / (2.0 * x * x);
    default: return NAN;
    }
}

// LDS-resident views of the model constants: uniform reads of ModelDev in the per-parameter loops would be global loads
// with ~1 µs dependent latency each (the compiler cannot prove them invariant), the views make them ds_reads.
struct ModelView {
    int d;
    const int *fixed, *prior_family;
    const double *lo, *hi, *prior_a, *prior_b, *prior_k;
};
struct LikView {
    int family;
    const double *par;
    double c0;
    const double *data;
    long long rows, cols;
    const double *aux;
    long long aux_rows, aux_cols;
};
constexpr int LIK_LDS_CAP = 768;          // doubles of likelihood data + regressors staged in LDS

template <class M, class Th>
__device__ inline bool in_bounds(const M &m, Th th) {
    bool ok = true;
#pragma unroll 4
    for (int k = 0; k < m.d; ++k) {
        const double x = th(k);
        ok = ok && (m.lo[k] <= x && x <= m.hi[k]);
    }
    return ok;
}

template <class M, class Th>
__device__ inline double logprior(const M &m, Th th) {
    double s = 0.0;
#pragma unroll 4
    for (int k = 0; k < m.d; ++k)
        if (!m.fixed[k]) s += prior_logpdf(m.prior_family[k], m.prior_a[k], m.prior_b[k], m.prior_k[k], th(k));
    return s;
}

// Compile-time-d variants: fully unrolled so that th(k) can index a register array.
template <int D, class M, class Th>
__device__ inline bool in_bounds_s(const M &m, Th th) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double x = th(k);
        ok = ok && (m.lo[k] <= x && x <= m.hi[k]);
    }
    return ok;
}
// Normal and Uniform priors (what the reference's examples and tests use) are evaluated branch-free inside the unrolled
// loop; the other families go through ONE rolled copy of prior_logpdf (selected by `has_other`), otherwise ten inlined
// copies of six log()-heavy cases bloat the kernel by ~60 KB of code and the wavefronts stall on instruction fetch.
template <int D, class M, class Th>
__device__ inline double logprior_s(const M &m, Th th, int has_other) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const int fam = m.prior_family[k];
        const double a = m.prior_a[k], b = m.prior_b[k], kk = m.prior_k[k], x = th(k);
        const double z = (x - a) / b;
        const double ln = -(z * z + LOG2PI) / 2.0 - kk;
        const double lu = (a <= x && x <= b) ? kk : SMCMI_NEG_INF;
        const double v = (fam == SMCMI_PRIOR_NORMAL) ? ln : lu;
        s += (m.fixed[k] || fam > SMCMI_PRIOR_UNIFORM) ? 0.0 : v;
    }
    if (has_other) {
#pragma nounroll
        for (int k = 0; k < D; ++k) {
            const int fam = m.prior_family[k];
            if (fam > SMCMI_PRIOR_UNIFORM && !m.fixed[k]) {
                double x = 0.0;
#pragma unroll
                for (int q = 0; q < D; ++q) x = (q == k) ? th(q) : x;
                s += prior_logpdf(fam, m.prior_a[k], m.prior_b[k], m.prior_k[k], x);
            }
        }
    }
    return s;
}

// host: family constant (same expression as the oracle so both sides share the libm value)
inline double lik_const_host(int family, const double *par, int d) {
    if (family == SMCMI_LIK_GAUSS_ISO) return -0.5 * (double)d * log(2.0 * M_PI * par[0] * par[0]);
    return 0.0;
}

// Linear-Gaussian state-space likelihood by the Kalman filter (build-defined model, no reference source).
// n_s = 8 states, n_y = 3 observables, n_r = 3 shocks, d = 13 parameters: th[0..7] = ρ, th[8..10] = shock std σ, th[11] =
// measurement std σ_e, th[12] = measurement mean μ.  x_t = Tm x_{t-1} + Rm ε_t, ε ~ N(0, diag σ²); y_t = μ + Z x_t + u_t,
// u ~ N(0, σ_e² I); Tm = diag(ρ) + κ C.  aux = [C (8x8) | Rm (8x3) | Z (3x8)] row-major, par0 = κ, y = 3 x T column-major.
// x_0 = 0, P_0 = I.  Innovations form with a 3x3 Cholesky of F_t; -Inf when F_t is not positive definite.
//
// One thread = one particle; one time step is fully unrolled so that every index is a compile-time constant and the whole
// filter state lives in registers: the covariance as a packed symmetric upper triangle (36 doubles, two copies), one row of
// Tm P at a time, P Z' (24) and its triangular solve (24).  C, Rm, Z are uniform across the wavefront (scalar loads); Tm and
// R Q R' are never stored: Tm[i][k] = κ C[i][k] (+ ρ_i on the diagonal) and the 36 entries of R Q R' are rebuilt from the three
// σ² each step (cheaper than 72 more registers).  The wave-uniform products κ C[i][k] and Rm[i][m] Rm[j][m] are formed once on
// the host when the likelihood is set (smcmi_set_likelihood appends them to the structure block: KALMAN_AUX_*), so they
// arrive as scalar operands instead of costing ~800 FP64 multiplies per filter step and lane - same roundings, same bits.
// Out of line so the callers' other likelihood families stay small.
constexpr int KALMAN_AUX_USER = 112;               // [C 8x8 | Rm 8x3 | Z 3x8] as the caller hands them over
constexpr int KALMAN_AUX_KC = KALMAN_AUX_USER;     // κ C, 64 doubles
constexpr int KALMAN_AUX_RR = KALMAN_AUX_KC + 64;  // Rm[i][m] Rm[j][m], (i <= j packed like P) x 3
constexpr int KALMAN_AUX_TOTAL = KALMAN_AUX_RR + 36 * 3;
__host__ __device__ constexpr int ksym(int i, int j) { return i <= j ? i * 8 - i * (i - 1) / 2 + (j - i) : j * 8 - j * (j - 1) / 2 + (i - j); }

// nt_mid: also report the log-likelihood of the first nt_mid observations (0 < nt_mid <= nt; a filter started from the same x_0, P_0
// on a prefix of the data performs exactly the first nt_mid steps of this one, so the value is bit for bit what a separate pass
// over the prefix returns) - the tempered update's old_loglikelihood when the old vintage is a prefix of the new one.
struct KalmanLL { double ll, ll_mid; };
__device__ __attribute__((noinline)) static KalmanLL kalman_lgss2(const double *thv, const double *ydat, long long nt, long long nt_mid, const double *aux, double kappa) {
SMCMI_FP_CONTRACT
    // The structure block and the data are the same for every lane, but an out-of-line function receives its pointers in VGPRs and
    // would fetch them with vector loads (105 flat loads and a dozen full waits per filter step): pin the addresses to SGPRs and
    // to the constant address space, so the values arrive through the scalar cache as SGPR operands of the FMAs.
    using cdp = const double __attribute__((address_space(4))) *;
    auto uniform_ptr = [](const double *p) -> cdp {
        const unsigned long long a = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (cdp)(((unsigned long long)hi << 32) | lo);
    };
    const cdp Zm = uniform_ptr(aux + 88), kC = uniform_ptr(aux + KALMAN_AUX_KC), RR = uniform_ptr(aux + KALMAN_AUX_RR), yd = uniform_ptr(ydat);
    nt = (long long)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)nt >> 32)) << 32) |
                     __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)nt));
    nt_mid = (long long)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)nt_mid >> 32)) << 32) |
                         __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)nt_mid));
    (void)kappa;
    double rho[8], s2[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) rho[i] = thv[i];
#pragma unroll
    for (int m = 0; m < 3; ++m) s2[m] = thv[8 + m] * thv[8 + m];
    const double se2 = thv[11] * thv[11], mu = thv[12];
    double P[36], Pn[36], x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = 0.0;
#pragma unroll
        for (int j = i; j < 8; ++j) P[ksym(i, j)] = (i == j) ? 1.0 : 0.0;
    }
    double ll = 0.0, ll_mid = SMCMI_NEG_INF;
#pragma nounroll
    for (long long t = 0; t < nt; ++t) {
        // (the 196 structure values are loop-invariant; the compiler hoists their scalar loads and parks what does not fit the SGPR
        // file in VGPR lanes - a v_readlane per use.  Reloading them every step instead was measured: 603 µs against 576 µs.)
        double xp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double s = rho[i] * x[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += kC[i * 8 + j] * x[j];
            xp[i] = s;
        }
        // P_{t|t-1} = Tm P Tm' + R Q R', upper triangle, one row of Tm P at a time
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double tp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double s = rho[i] * P[ksym(i, j)];
#pragma unroll
                for (int k = 0; k < 8; ++k) s += kC[i * 8 + k] * P[ksym(k, j)];
                tp[j] = s;
            }
#pragma unroll
            for (int j = i; j < 8; ++j) {
                double s = RR[ksym(i, j) * 3 + 0] * s2[0] + RR[ksym(i, j) * 3 + 1] * s2[1] + RR[ksym(i, j) * 3 + 2] * s2[2];
                s += tp[j] * rho[j];
#pragma unroll
                for (int k = 0; k < 8; ++k) s += tp[k] * kC[j * 8 + k];
                Pn[ksym(i, j)] = s;
            }
        }
        double v[3], PZ[24], F[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += Zm[a * 8 + j] * xp[j];
            v[a] = yd[a + 3 * t] - mu - s;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += Pn[ksym(i, j)] * Zm[a * 8 + j];
                PZ[i * 3 + a] = s;
            }
        // F = Z P Z' + σ_e² I (lower triangle: F00 F10 F11 F20 F21 F22)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                double s = (a == b) ? se2 : 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += Zm[a * 8 + i] * PZ[i * 3 + b];
                F[a * (a + 1) / 2 + b] = s;
            }
        if (!(F[0] > 0.0)) return KalmanLL{SMCMI_NEG_INF, ll_mid};
        const double l00 = sqrt(F[0]), l10 = F[1] / l00, l20 = F[3] / l00;
        const double p11 = F[2] - l10 * l10;
        if (!(p11 > 0.0)) return KalmanLL{SMCMI_NEG_INF, ll_mid};
        const double l11 = sqrt(p11), l21 = (F[4] - l20 * l10) / l11;
        const double p22 = F[5] - l20 * l20 - l21 * l21;
        if (!(p22 > 0.0)) return KalmanLL{SMCMI_NEG_INF, ll_mid};
        const double l22 = sqrt(p22);
        const double w0 = v[0] / l00, w1 = (v[1] - l10 * w0) / l11, w2 = (v[2] - l20 * w0 - l21 * w1) / l22;
        ll += -1.5 * log(2.0 * M_PI) - log(l00 * l11 * l22) - 0.5 * (w0 * w0 + w1 * w1 + w2 * w2);
        if (t + 1 == nt_mid) ll_mid = ll;                       // (wave-uniform)
        const double u2 = w2 / l22, u1 = (w1 - l21 * u2) / l11, u0 = (w0 - l10 * u1 - l20 * u2) / l00;
        const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
        double G[24];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x[i] = xp[i] + (PZ[i * 3 + 0] * u0 + PZ[i * 3 + 1] * u1 + PZ[i * 3 + 2] * u2);
            const double g0 = PZ[i * 3 + 0] * i00, g1 = (PZ[i * 3 + 1] - l10 * g0) * i11;
            G[i * 3 + 0] = g0; G[i * 3 + 1] = g1; G[i * 3 + 2] = (PZ[i * 3 + 2] - l20 * g0 - l21 * g1) * i22;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = i; j < 8; ++j)
                P[ksym(i, j)] = Pn[ksym(i, j)] - (G[i * 3 + 0] * G[j * 3 + 0] + G[i * 3 + 1] * G[j * 3 + 1] + G[i * 3 + 2] * G[j * 3 + 2]);
    }
    return KalmanLL{ll, ll_mid};
}
__device__ inline double kalman_lgss(const double *thv, const double *ydat, long long nt, const double *aux, double kappa) {
    return kalman_lgss2(thv, ydat, nt, 0, aux, kappa).ll;
}

template <class L, class Th>
__device__ inline double loglik(const L &l, int d, Th th) {
    switch (l.family) {
    case SMCMI_LIK_GAUSS_ISO: {
        const double sig = l.par[0];
        double acc = 0.0;
#pragma unroll 4
        for (int k = 0; k < d; ++k) { const double e = th(k) - l.data[k]; acc += e * e; }
        return l.c0 - acc / (2.0 * sig * sig);
    }
    case SMCMI_LIK_LINREG: {  // examples/regression_model/estimate_regression.jl:46-53, data = [y X]
        const long long n = l.rows;
        const double *y = l.data, *X = l.data + n;
        const double s2 = l.par[0], Nn = (double)n, a = th(0), b = th(1);
        const double term1 = -(Nn / 2.0) * log(2.0 * M_PI) - (Nn / 2.0) * log(s2);
        double dot = 0.0;
#pragma unroll 4
        for (long long t = 0; t < n; ++t) { const double e = y[t] - a - b * X[t]; dot += e * e; }
        return term1 - (1.0 / (2.0 * s2)) * dot;
    }
    case SMCMI_LIK_LINMODEL3: {  // test/modelsetup.jl:119-138 loglik_fn
        const long long T = l.cols;
        double a[3], b[3], inv[3], det = 1.0;
        bool singular = false;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[i] = th(3 * i); b[i] = th(3 * i + 1);
            const double s = th(3 * i + 2), v = s * s;
            singular = singular || (v == 0.0);
            det *= v; inv[i] = 1.0 / v;
        }
        if (singular) return SMCMI_NEG_INF;
        const double term1 = -3.0 / 2.0 * log(2.0 * M_PI) - 1.0 / 2.0 * log(det);
        double lp = 0.0;
        for (long long t = 0; t < T; ++t) {
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double e = l.data[i + 3 * t] - a[i] - b[i] * l.aux[i + l.aux_rows * t];
                q += e * (inv[i] * e);
            }
            lp += term1 - 1.0 / 2.0 * q;
        }
        return lp;
    }
    case SMCMI_LIK_LGSS_KALMAN: {  // config 5: dense linear state-space model, Kalman filter per particle
        if (d != 13 || l.rows != 3 || l.aux_rows * l.aux_cols < KALMAN_AUX_TOTAL) return NAN;
        double thv[13];
        for (int k = 0; k < 13; ++k) thv[k] = th(k);
        return kalman_lgss(thv, l.data, l.cols, l.aux, l.par[0]);
    }
    case SMCMI_LIK_CAPM_LITERAL: {  // examples/capm_model/estimate_capm.jl:52-70 as written (quirk Q12)
        const long long T = l.cols;
        double a[3], inv[3], det = 1.0;
        bool singular = false;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[i] = th(3 * i);
            const double s = th(3 * i + 2), v = s * s;
            singular = singular || (v == 0.0);
            det *= v; inv[i] = 1.0 / v;
        }
        if (singular) return SMCMI_NEG_INF;
        const double term1 = -3.0 / 2.0 * log(2.0 * M_PI) - 1.0 / 2.0 * log(det);
        double S = 0.0, lp = 0.0;
        for (long long t = 0; t < T; ++t) {
            const double mk = l.aux[l.aux_rows * t];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double e = l.data[i + 3 * t] - a[i] - a[i] * mk;
                S += e * (inv[i] * e);
            }
        }
        for (long long t = 0; t < T; ++t) lp += term1 - 1.0 / 2.0 * S;
        return lp;
    }
    default: return NAN;
    }
}



template <int D, class L, class Th>
__device__ inline double loglik_s(const L &l, Th th) {
    switch (l.family) {
    case SMCMI_LIK_GAUSS_ISO: {
        const double sig = l.par[0];
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { const double e = th(k) - l.data[k]; acc += e * e; }
        return l.c0 - acc / (2.0 * sig * sig);
    }
    case SMCMI_LIK_LINREG: if constexpr (D == 2) {  // examples/regression_model/estimate_regression.jl:46-53, data = [y X]
        const long long n = l.rows;
        const double *y = l.data, *X = l.data + n;
        const double s2 = l.par[0], Nn = (double)n, a = th(0), b = th(1);
        const double term1 = -(Nn / 2.0) * log(2.0 * M_PI) - (Nn / 2.0) * log(s2);
        double dot = 0.0;
#pragma unroll 4
        for (long long t = 0; t < n; ++t) { const double e = y[t] - a - b * X[t]; dot += e * e; }
        return term1 - (1.0 / (2.0 * s2)) * dot;
    } else return NAN;
    case SMCMI_LIK_LINMODEL3: if constexpr (D == 9) {  // test/modelsetup.jl:119-138 loglik_fn
        const long long T = l.cols;
        double a[3], b[3], inv[3], det = 1.0;
        bool singular = false;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[i] = th(3 * i); b[i] = th(3 * i + 1);
            const double s = th(3 * i + 2), v = s * s;
            singular = singular || (v == 0.0);
            det *= v; inv[i] = 1.0 / v;
        }
        if (singular) return SMCMI_NEG_INF;
        const double term1 = -3.0 / 2.0 * log(2.0 * M_PI) - 1.0 / 2.0 * log(det);
        double lp = 0.0;
        for (long long t = 0; t < T; ++t) {
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double e = l.data[i + 3 * t] - a[i] - b[i] * l.aux[i + l.aux_rows * t];
                q += e * (inv[i] * e);
            }
            lp += term1 - 1.0 / 2.0 * q;
        }
        return lp;
    } else return NAN;
    case SMCMI_LIK_CAPM_LITERAL: if constexpr (D == 9) {  // examples/capm_model/estimate_capm.jl:52-70 as written (quirk Q12)
        const long long T = l.cols;
        double a[3], inv[3], det = 1.0;
        bool singular = false;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[i] = th(3 * i);
            const double s = th(3 * i + 2), v = s * s;
            singular = singular || (v == 0.0);
            det *= v; inv[i] = 1.0 / v;
        }
        if (singular) return SMCMI_NEG_INF;
        const double term1 = -3.0 / 2.0 * log(2.0 * M_PI) - 1.0 / 2.0 * log(det);
        double S = 0.0, lp = 0.0;
        // the observations are the same for every lane: read through the constant address space they arrive as scalar operands (the
        // register kernels hand this family the global pointers, not an LDS copy)
        using cdp = const double __attribute__((address_space(4))) *;
        const cdp dat = (cdp)(unsigned long long)l.data, mkt = (cdp)(unsigned long long)l.aux;
        const long long ar = l.aux_rows;
#pragma unroll 4
        for (long long t = 0; t < T; ++t) {
            const double mk = mkt[ar * t];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double e = dat[i + 3 * t] - a[i] - a[i] * mk;
                S += e * (inv[i] * e);
            }
        }
        for (long long t = 0; t < T; ++t) lp += term1 - 1.0 / 2.0 * S;
        return lp;
    } else return NAN;
    default: return NAN;
    }
}



}  // namespace smcmi
