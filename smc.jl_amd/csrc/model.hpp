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
#include <type_traits>

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
    nt = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)nt >> 32)) << 32) |
                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)nt));
    nt_mid = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)nt_mid >> 32)) << 32) |
                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)nt_mid));
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

// ---- the same filter with FOUR lanes per particle (one DPP quad): lane q owns rows 2q, 2q+1 of the covariance.
// One thread per particle keeps ~460 registers alive and leaves a cloud of 12 500 particles (BASELINE config 5 on 4 GPUs) on a fifth of
// the SIMDs; here a wavefront carries 16 particles, a lane ~130 registers, and the same cloud fills 782 wavefronts.  How the rows
// stay local:
//   W  = P Tm'            row r of W needs row r of P, ρ (replicated) and the wave-uniform κC                 - local
//   TP = Tm P = W'        (P symmetric)  one 8x8 transpose per step through the particle's 576-byte LDS slot (16-byte accesses,
//                         conflict-free layout: element (k, c) at (k>>1) 144 + (k&1) 64 + 8 c)
//   Pn = TP Tm' + R Q R'  local again; R Q R' rows are loop invariants (16 registers instead of 48 multiplies per step)
//   PZ = Pn Z'            local
//   F = Z PZ + σ_e² I, Z x_p   two-row partial sums, one quad all-reduce of 9 values (DPP quad_perm: every lane gets the same bits)
//   3x3 factorisation     replicated in the quad; reciprocal square roots (hardware estimate + two Newton steps) instead of
//                         sqrt + 9 divisions, and the determinant as a running product with its exponent split off (one log per
//                         evaluation instead of one per step)
//   P  = Pn - G G'        G = PZ L^-T: own rows local, the other six rows (and the other six entries of x) by quad broadcasts
// The 88 wave-uniform structure values (κC, Z) do not fit the scalar registers next to everything else (the compiler parks them in
// VGPR lanes and pays two v_readlane per use: 380 of 1 250 instructions per step when written with scalar operands).  Here they sit
// in SIX vector registers - value c in lane c mod 16 of every 16-lane row of register c / 16 - and reach the FMA through its own
// DPP operand (`v_fmac_f64_dpp ... row_newbcast:n`, the one DPP control gfx90a+ has for 64-bit operations): no instruction, no
// scalar register per use.  A DPP source lane must be active, so ALL 64 LANES MUST CALL THIS TOGETHER (quads without a valid
// parameter vector pass any finite numbers and ignore the result).
// P is not symmetrised (the asymmetric part is rounding noise that Tm contracts).  Values agree with kalman_lgss2 to ~1e-13
// relative, not bit for bit (different summation orders).  xslot: this particle's LDS slot (KALMAN4_SLOT_BYTES, 16-byte aligned).
constexpr int KALMAN4_SLOT_BYTES = 576;
typedef __attribute__((address_space(3))) char *lds_bytes;
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2f64 *lds_v2f64;
template <int CTRL>
__device__ inline double quad_perm_f64(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double quad_allsum(double x) {
    x += quad_perm_f64<0xB1>(x);          // [1,0,3,2]
    x += quad_perm_f64<0x4E>(x);          // [2,3,0,1]
    return x;
}
__device__ inline double rsqrt_nr(double a) {
    double y = __builtin_amdgcn_rsq(a);
    const double h0 = 0.5 * a;
    y = y * (1.5 - h0 * y * y);
    y = y * (1.5 - h0 * y * y);
    return y;
}
// acc += (structure value C) * x, the value taken from lane C % 16 of this lane's row of cst[C / 16]
template <int C, int NC>
__device__ inline void fmac_row_const(double &acc, const double (&cst)[NC], double x) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(cst[C / 16]), "v"(x), "n"(C % 16));
}
template <int J, int K = 0>
struct RowDot8 {                 // acc += Σ_k (structure value BASE + 8 J + k) * x[k]   (SKIPD: without k = J)
    template <int BASE, bool SKIPD>
    __device__ static inline void run(double &acc, const double (&cst)[6], const double (&x)[8]) {
        if constexpr (!(SKIPD && K == J)) fmac_row_const<BASE + 8 * J + K>(acc, cst, x[K]);
        if constexpr (K < 7) RowDot8<J, K + 1>::template run<BASE, SKIPD>(acc, cst, x);
    }
};
template <int BASE, int J = 0>
struct MatRows8 {                // out[j] = init[j] + Σ_k value(BASE + 8 j + k) x[k], j < NJ
    template <int NJ, bool SKIPD = false>
    __device__ static inline void run(double *out, const double (&cst)[6], const double (&x)[8]) {
        double acc = out[J];
        RowDot8<J>::template run<BASE, SKIPD>(acc, cst, x);
        out[J] = acc;
        if constexpr (J + 1 < NJ) MatRows8<BASE, J + 1>::template run<NJ, SKIPD>(out, cst, x);
    }
};
// log-likelihood of the first `steps` observations from the running sums (one expression, never contracted: the value at the old
// vintage's last period must be bit for bit what a separate pass over that prefix returns)
__device__ inline double kalman_quad_value(long long steps, double detm, int dete, double quad_acc, bool bad) {
#pragma clang fp contract(off)
    // A filter that lost positive definiteness keeps running on NaN / Inf (its lanes cannot leave the wavefront's lockstep).  None of
    // that may reach log(): its argument is clamped into the range a live filter's mantissa has (fmax / fmin drop a NaN) - no branch
    // around the call (a divergent region inside the filter loop next to the DPP operands was measured to end in memory faults).
    const double ln2 = 0.693147180559945309417232121458;
    const double dm = fmin(fmax(detm, 0.25), 1.0);
    const double a = -1.5 * (double)steps * LOG2PI;
    const double b = log(dm) + (double)dete * ln2;
    const double v = (a + b) - 0.5 * quad_acc;
    return (bad || !(detm >= 0.25) || !(detm <= 1.0)) ? SMCMI_NEG_INF : v;
}
__device__ __forceinline__ static KalmanLL kalman_lgss_quad(const double *thv, const double *ydat, long long nt, long long nt_mid, const double *aux,
                                                                     lds_bytes xslot, int q) {
SMCMI_FP_CONTRACT
    using cdp = const double __attribute__((address_space(4))) *;
    auto uniform_ptr = [](const double *p) -> cdp {
        const unsigned long long a = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (cdp)(((unsigned long long)hi << 32) | lo);
    };
    const cdp yd = uniform_ptr(ydat);
    const double *RRg = aux + KALMAN_AUX_RR, *kCg = aux + KALMAN_AUX_KC, *Zg = aux + 88;
    nt = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)nt >> 32)) << 32) |
                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)nt));
    nt_mid = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)nt_mid >> 32)) << 32) |
                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)nt_mid));
    const int r0 = 2 * q, l16 = (int)(threadIdx.x & 15);
    double cst[6];               // structure values 0..63 = κC (row-major), 64..87 = Z (row-major); lane c % 16 of register c / 16
#pragma unroll
    for (int g = 0; g < 4; ++g) cst[g] = kCg[16 * g + l16];
    cst[4] = Zg[l16];
    cst[5] = Zg[16 + (l16 & 7)];
    double dT[8], s2[3];          // dT: diagonal of Tm = ρ_j + κC[j][j] (one operand instead of a product and an FMA per use)
#pragma unroll
    for (int i = 0; i < 8; ++i) dT[i] = thv[i] + kCg[9 * i];
#pragma unroll
    for (int m = 0; m < 3; ++m) s2[m] = thv[8 + m] * thv[8 + m];
    const double se2 = thv[11] * thv[11], mu = thv[12];
    double P[2][8], Qt[2][8], kr[2][8], zc[3][2], x_own[2], xg[8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int i = r0 + s;
        x_own[s] = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = ksym(i, j);
            Qt[s][j] = RRg[p * 3 + 0] * s2[0] + RRg[p * 3 + 1] * s2[1] + RRg[p * 3 + 2] * s2[2];
            kr[s][j] = (i == j) ? dT[j] : kCg[i * 8 + j];               // own rows of Tm = κC + diag ρ
            P[s][j] = (i == j) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) zc[a][s] = Zg[a * 8 + i];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) xg[j] = 0.0;
    const lds_v2f64 wrow[2] = {(lds_v2f64)(xslot + q * 144), (lds_v2f64)(xslot + q * 144 + 64)};
    const lds_bytes rcol = xslot + q * 16;
    double quad_acc = 0.0, detm = 1.0, mid_detm = 1.0, mid_quad = 0.0;
    int dete = 0, mid_dete = 0;
    bool bad = false, mid_bad = true;
    double ll_mid = SMCMI_NEG_INF;
    asm volatile("s_nop 4");     // (the loads above and the first DPP read of cst are far apart anyway)
#pragma nounroll
    for (long long t = 0; t < nt; ++t) {
        double xp[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            double v = kr[s][0] * xg[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) v += kr[s][j] * xg[j];
            xp[s] = v;
        }
        // W = P Tm' (own rows) -> LDS
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            double w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = dT[j] * P[s][j];
            MatRows8<0>::run<8, true>(w, cst, P[s]);
#pragma unroll
            for (int c = 0; c < 4; ++c) { v2f64 pr; pr.x = w[2 * c]; pr.y = w[2 * c + 1]; wrow[s][c] = pr; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double TP[2][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const v2f64 pr = *(lds_v2f64)(rcol + (k >> 1) * 144 + (k & 1) * 64);
            TP[0][k] = pr.x; TP[1][k] = pr.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double Pn[2][8], PZ[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) Pn[s][j] = Qt[s][j] + TP[s][j] * dT[j];
            MatRows8<0>::run<8, true>(Pn[s], cst, TP[s]);
            PZ[s][0] = PZ[s][1] = PZ[s][2] = 0.0;
            MatRows8<64>::run<3>(PZ[s], cst, Pn[s]);
        }
        // nine two-row partial sums, totalled over the quad together (a DPP move must not follow the write of its source directly:
        // batching the nine values keeps the hazard's wait states filled with the other eight)
        double red[9];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[a] = zc[a][0] * xp[0] + zc[a][1] * xp[1];
#pragma unroll
            for (int b = 0; b <= a; ++b) red[3 + a * (a + 1) / 2 + b] = zc[a][0] * PZ[0][b] + zc[a][1] * PZ[1][b];
        }
        {
            double m[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) m[i] = quad_perm_f64<0xB1>(red[i]);          // [1,0,3,2]
#pragma unroll
            for (int i = 0; i < 9; ++i) red[i] += m[i];
#pragma unroll
            for (int i = 0; i < 9; ++i) m[i] = quad_perm_f64<0x4E>(red[i]);          // [2,3,0,1]
#pragma unroll
            for (int i = 0; i < 9; ++i) red[i] += m[i];
        }
        double zx[3], F[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            zx[a] = red[a];
#pragma unroll
            for (int b = 0; b <= a; ++b) F[a * (a + 1) / 2 + b] = red[3 + a * (a + 1) / 2 + b] + ((a == b) ? se2 : 0.0);
        }
        const double v0 = yd[0 + 3 * t] - mu - zx[0], v1 = yd[1 + 3 * t] - mu - zx[1], v2 = yd[2 + 3 * t] - mu - zx[2];
        bad = bad || !(F[0] > 0.0);
        const double i00 = rsqrt_nr(F[0]), l10 = F[1] * i00, l20 = F[3] * i00;
        const double p11 = F[2] - l10 * l10;
        bad = bad || !(p11 > 0.0);
        const double i11 = rsqrt_nr(p11), l21 = (F[4] - l20 * l10) * i11;
        const double p22 = F[5] - l20 * l20 - l21 * l21;
        bad = bad || !(p22 > 0.0);
        const double i22 = rsqrt_nr(p22);
        const double w0 = v0 * i00, w1 = (v1 - l10 * w0) * i11, w2 = (v2 - l20 * w0 - l21 * w1) * i22;
        quad_acc += w0 * w0 + w1 * w1 + w2 * w2;
        detm *= (i00 * i11) * i22;                                  // 1 / (l00 l11 l22)
        dete += __builtin_amdgcn_frexp_exp(detm);
        detm = __builtin_amdgcn_frexp_mant(detm);
        if (t + 1 == nt_mid) { mid_detm = detm; mid_dete = dete; mid_quad = quad_acc; mid_bad = bad; }      // (wave-uniform; evaluated after the loop)
        const double u2 = w2 * i22, u1 = (w1 - l21 * u2) * i11, u0 = (w0 - l10 * u1 - l20 * u2) * i00;
        double g[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            x_own[s] = xp[s] + (PZ[s][0] * u0 + PZ[s][1] * u1 + PZ[s][2] * u2);
            g[s][0] = PZ[s][0] * i00;
            g[s][1] = (PZ[s][1] - l10 * g[s][0]) * i11;
            g[s][2] = (PZ[s][2] - l20 * g[s][0] - l21 * g[s][1]) * i22;
        }
        // rows (G[j][0..2], x[j]) of all eight states: quad broadcasts.  (Through LDS - 4 writes + 16 reads, no VALU slot - was measured:
        // 231 µs against 210 µs per mutation at one wavefront per SIMD, the extra round trip costs more than the 64 DPP moves.)
        double G[8][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            xg[0 + s] = quad_perm_f64<0x00>(x_own[s]); xg[2 + s] = quad_perm_f64<0x55>(x_own[s]);
            xg[4 + s] = quad_perm_f64<0xAA>(x_own[s]); xg[6 + s] = quad_perm_f64<0xFF>(x_own[s]);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                G[0 + s][a] = quad_perm_f64<0x00>(g[s][a]); G[2 + s][a] = quad_perm_f64<0x55>(g[s][a]);
                G[4 + s][a] = quad_perm_f64<0xAA>(g[s][a]); G[6 + s][a] = quad_perm_f64<0xFF>(g[s][a]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double v = Pn[s][j];
                v -= g[s][0] * G[j][0]; v -= g[s][1] * G[j][1]; v -= g[s][2] * G[j][2];
                P[s][j] = v;
            }
    }
    const double ll = kalman_quad_value(nt, detm, dete, quad_acc, bad);
    if (nt_mid > 0 && nt_mid <= nt) ll_mid = kalman_quad_value(nt_mid, mid_detm, mid_dete, mid_quad, mid_bad);      // (wave-uniform)
    return KalmanLL{ll, ll_mid};
}

// ---- one thread per particle again, for clouds too large for the lane-split form - but with what the lane-split filter taught:
// the 196 wave-uniform structure values (Z, κC, Rm Rm') sit in THIRTEEN vector registers (value c in lane c mod 16 of every 16-lane
// row of register c / 16) and reach the FMAs through their DPP operand - kalman_lgss2 keeps them as scalar operands, which do not fit
// the SGPR file: the compiler parks them in VGPR lanes and pays ~400 v_readlane per step (17 % of its instructions); the 3x3
// factorisation uses reciprocal square roots (no sqrt, no division), the determinant a running product (one log per evaluation).
// A DPP source lane must be active, whatever lanes the caller arrives with: the function widens EXEC itself and restores it before
// returning (see below) - any WAVE-UNIFORM call site will do, the arguments other than θ must be the same in every lane.  Callers
// inside per-particle loops that end early (the initial draw's redraw loop) keep kalman_lgss2.  Same algorithm, different summation
// order in places: values agree to ~1e-12.
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(static_cast<F &&>(f));
    }
}
#define SMCMI_CI(name, T) constexpr int name = decltype(T)::value
struct KalmanTheta { double v[13]; };
__device__ __attribute__((noinline)) static KalmanLL kalman_lgss_wave(const KalmanTheta th_in, const double *ydat, long long nt, long long nt_mid, const double *aux) {
SMCMI_FP_CONTRACT
    using cdp = const double __attribute__((address_space(4))) *;
    // (readfirstlane returns int: without the casts the low half would be SIGN-extended over the high half - a fault whenever bit 31 of
    // an address is set)
    auto uniform_u64 = [](unsigned long long a) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
    };
    unsigned long long u_yd = uniform_u64((unsigned long long)ydat), u_aux = uniform_u64((unsigned long long)aux);
    unsigned long long u_nt = uniform_u64((unsigned long long)nt), u_mid = uniform_u64((unsigned long long)nt_mid);
    double thv[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) thv[k] = th_in.v[k];
    // THE WHOLE WAVEFRONT RUNS THE FILTER, whatever lanes the caller arrives with: a DPP operand needs its source lane (lanes 0..15 of each
    // 16-lane row hold the structure values) active, and the callers' control flow around a particle's proposal is not uniform in
    // general (measured: the mutation kernel reaches this call with only the lanes that own a particle; their results were garbage
    // whenever the last block of a cloud was partly empty).  EXEC is widened here and restored before the return; lanes that were not
    // active run on ρ = 0, σ = 0.5, μ = 0 (finite values: nothing of them survives the restore).  Everything per lane that is read
    // below comes from uniform values and the lane number, never from a register an inactive lane may hold garbage in.
    // (one statement: the incoming mask is saved, EXEC widened, and the uniform values pass THROUGH it - they are read from the first
    // lane that was active, so they must exist before the mask changes and everything below must use the copies that come out)
    unsigned long long exec_in;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, -1" : "=&s"(exec_in), "+s"(u_yd), "+s"(u_aux), "+s"(u_nt), "+s"(u_mid) : : "memory");
    const cdp yd = (cdp)u_yd;
    const double *auxu = (const double *)u_aux;
    nt = (long long)u_nt;
    nt_mid = (long long)u_mid;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool was_active = (exec_in >> lane) & 1ull;
#pragma unroll
    for (int k = 0; k < 13; ++k) thv[k] = was_active ? thv[k] : ((k >= 8 && k < 12) ? 0.5 : 0.0);
    // structure values as they lie from aux + 88 on: Z (24) | κC (64) | Rm Rm' (36 x 3)
    constexpr int CZ = 0, CK = 24, CR = 88, NCST = 13;
    const int l16 = lane & 15;
    double cst[NCST];
#pragma unroll
    for (int g = 0; g < NCST; ++g) { const int c = 16 * g + l16; cst[g] = auxu[88 + (c < 196 ? c : 195)]; }
    double rho[8], s2[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) rho[i] = thv[i];
#pragma unroll
    for (int m = 0; m < 3; ++m) s2[m] = thv[8 + m] * thv[8 + m];
    const double se2 = thv[11] * thv[11], mu = thv[12];
    double dT[8];                 // diagonal of Tm = ρ_i + κC[i][i]: one operand instead of a product and an FMA per use
#pragma unroll
    for (int i = 0; i < 8; ++i) dT[i] = rho[i] + auxu[88 + CK + 9 * i];
    double P[36], Pn[36], x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = 0.0;
#pragma unroll
        for (int j = i; j < 8; ++j) P[ksym(i, j)] = (i == j) ? 1.0 : 0.0;
    }
    double quad_acc = 0.0, detm = 1.0, ll_mid = SMCMI_NEG_INF, mid_detm = 1.0, mid_quad = 0.0;
    int dete = 0, mid_dete = 0;
    bool bad = false, mid_bad = true;
    asm volatile("s_nop 4");
#pragma nounroll
    for (long long t = 0; t < nt; ++t) {
        // (the reduction index runs OUTERMOST everywhere: consecutive instructions then feed different accumulators - the scheduler
        // knows nothing about the latency of the inline-asm FMAs and leaves them in source order)
        double xp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xp[i] = dT[i] * x[i];
        static_for<8>([&](auto J) __attribute__((always_inline)) {
            SMCMI_CI(j, J);
            static_for<8>([&](auto I) __attribute__((always_inline)) { SMCMI_CI(i, I); if constexpr (i != j) fmac_row_const<CK + 8 * i + j>(xp[i], cst, x[j]); });
        });
        // P_{t|t-1} = Tm P Tm' + R Q R', upper triangle, one row of Tm P at a time
        static_for<8>([&](auto I) __attribute__((always_inline)) {
            SMCMI_CI(i, I);
            double tp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tp[j] = dT[i] * P[ksym(i, j)];
            static_for<8>([&](auto K) __attribute__((always_inline)) {
                SMCMI_CI(k, K);
                if constexpr (k != i)
                    static_for<8>([&](auto J) __attribute__((always_inline)) { SMCMI_CI(j, J); fmac_row_const<CK + 8 * i + k>(tp[j], cst, P[ksym(k, j)]); });
            });
            double pn[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pn[j] = tp[j] * dT[j];
            static_for<3>([&](auto M) __attribute__((always_inline)) {
                SMCMI_CI(m, M);
                static_for<8>([&](auto J) __attribute__((always_inline)) {
                    SMCMI_CI(j, J);
                    if constexpr (j >= i) fmac_row_const<CR + 3 * ksym(i, j) + m>(pn[j], cst, s2[m]);
                });
            });
            static_for<8>([&](auto K) __attribute__((always_inline)) {
                SMCMI_CI(k, K);
                static_for<8>([&](auto J) __attribute__((always_inline)) {
                    SMCMI_CI(j, J);
                    if constexpr (j >= i && k != j) fmac_row_const<CK + 8 * j + k>(pn[j], cst, tp[k]);
                });
            });
#pragma unroll
            for (int j = i; j < 8; ++j) Pn[ksym(i, j)] = pn[j];
        });
        double v[3], PZ[24], F[6];
        {
            double zx[3] = {0.0, 0.0, 0.0};
            static_for<8>([&](auto J) __attribute__((always_inline)) {
                SMCMI_CI(j, J);
                static_for<3>([&](auto A) __attribute__((always_inline)) { SMCMI_CI(a, A); fmac_row_const<CZ + 8 * a + j>(zx[a], cst, xp[j]); });
            });
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = yd[a + 3 * t] - mu - zx[a];
        }
#pragma unroll
        for (int e = 0; e < 24; ++e) PZ[e] = 0.0;
        static_for<8>([&](auto J) __attribute__((always_inline)) {
            SMCMI_CI(j, J);
            static_for<8>([&](auto I) __attribute__((always_inline)) {
                SMCMI_CI(i, I);
                static_for<3>([&](auto A) __attribute__((always_inline)) { SMCMI_CI(a, A); fmac_row_const<CZ + 8 * a + j>(PZ[i * 3 + a], cst, Pn[ksym(i, j)]); });
            });
        });
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b2 = 0; b2 <= a; ++b2) F[a * (a + 1) / 2 + b2] = (a == b2) ? se2 : 0.0;
        static_for<8>([&](auto I) __attribute__((always_inline)) {
            SMCMI_CI(i, I);
            static_for<3>([&](auto A) __attribute__((always_inline)) {
                SMCMI_CI(a, A);
                static_for<3>([&](auto B) __attribute__((always_inline)) {
                    SMCMI_CI(b2, B);
                    if constexpr (b2 <= a) fmac_row_const<CZ + 8 * a + i>(F[a * (a + 1) / 2 + b2], cst, PZ[i * 3 + b2]);
                });
            });
        });
        bad = bad || !(F[0] > 0.0);
        const double i00 = rsqrt_nr(F[0]), l10 = F[1] * i00, l20 = F[3] * i00;
        const double p11 = F[2] - l10 * l10;
        bad = bad || !(p11 > 0.0);
        const double i11 = rsqrt_nr(p11), l21 = (F[4] - l20 * l10) * i11;
        const double p22 = F[5] - l20 * l20 - l21 * l21;
        bad = bad || !(p22 > 0.0);
        const double i22 = rsqrt_nr(p22);
        const double w0 = v[0] * i00, w1 = (v[1] - l10 * w0) * i11, w2 = (v[2] - l20 * w0 - l21 * w1) * i22;
        quad_acc += w0 * w0 + w1 * w1 + w2 * w2;
        detm *= (i00 * i11) * i22;
        dete += __builtin_amdgcn_frexp_exp(detm);
        detm = __builtin_amdgcn_frexp_mant(detm);
        if (t + 1 == nt_mid) { mid_detm = detm; mid_dete = dete; mid_quad = quad_acc; mid_bad = bad; }      // (wave-uniform; evaluated after the loop)
        const double u2 = w2 * i22, u1 = (w1 - l21 * u2) * i11, u0 = (w0 - l10 * u1 - l20 * u2) * i00;
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
            for (int j = i; j < 8; ++j) {
                double s = Pn[ksym(i, j)];
                s -= G[i * 3 + 0] * G[j * 3 + 0]; s -= G[i * 3 + 1] * G[j * 3 + 1]; s -= G[i * 3 + 2] * G[j * 3 + 2];
                P[ksym(i, j)] = s;
            }
    }
    const double ll = kalman_quad_value(nt, detm, dete, quad_acc, bad);
    if (nt_mid > 0 && nt_mid <= nt) ll_mid = kalman_quad_value(nt_mid, mid_detm, mid_dete, mid_quad, mid_bad);      // (wave-uniform)
    asm volatile("s_mov_b64 exec, %0" : : "s"(exec_in) : "memory");
    return KalmanLL{ll, ll_mid};
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
