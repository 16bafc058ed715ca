/*
 * smc_oracle.c - CPU ORACLE (test infrastructure; see smc_oracle.h).  Plain FP64 C restatement of
 * the FRBNY-DSGE/SMC.jl hot path.  "ref:" comments cite /root/reference files (file:line).
 * Parity status: RNG-free pieces pinned by the reference's golden fixtures (tests/test_oracle_golden.py);
 * RNG streams are Philox and therefore unpinned against Julia's MersenneTwister.
 */
#include "smc_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdatomic.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXD 512
static const double LOG2PI = 1.8378770664093454835606594728112;
static const double TWO_PI = 6.283185307179586476925286766559;

static __thread char g_err[512];
const char *orc_last_error(void) { return g_err; }
static int fail(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return -1; }

/* ------------------------------------------------------------------ RNG contract */
void orc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                       uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static double u53(uint32_t hi, uint32_t lo) {
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * 0x1.0p-53;
}
void orc_uniform_pair(uint64_t seed, uint64_t pid, uint32_t stage, uint32_t tag, double *ua, double *ub) {
    uint32_t o[4];
    orc_philox4x32_10((uint32_t)pid, (uint32_t)(pid >> 32), stage, tag, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    *ua = u53(o[0], o[1]);
    *ub = u53(o[2], o[3]);
}
static void normal_pair(uint64_t seed, uint64_t pid, uint32_t stage, uint32_t tag, double *z0, double *z1) {
    double ua, ub;
    orc_uniform_pair(seed, pid, stage, tag, &ua, &ub);
    double r = sqrt(-2.0 * log(ua)), a = TWO_PI * ub;
    *z0 = r * cos(a);
    *z1 = r * sin(a);
}
#define TAG(purpose, t, q) (((uint32_t)(purpose) << 28) | (((uint32_t)(t) & 0xFFFFFu) << 8) | ((uint32_t)(q) & 0xFFu))
enum { P_MUT = 0, P_RES = 1, P_BLK = 2, P_INIT = 3 };

/* ------------------------------------------------------------------ small dense linear algebra */
static int chol_lower(const double *A, int n, double *L) { /* row-major n x n; returns -1 if not PD */
    memset(L, 0, sizeof(double) * n * n);
    for (int jx = 0; jx < n; ++jx) {
        double s = A[jx * n + jx];
        for (int k = 0; k < jx; ++k) s -= L[jx * n + k] * L[jx * n + k];
        if (!(s > 0.0)) return -1;
        double ljj = sqrt(s);
        L[jx * n + jx] = ljj;
        for (int ix = jx + 1; ix < n; ++ix) {
            double t = A[ix * n + jx];
            for (int k = 0; k < jx; ++k) t -= L[ix * n + k] * L[jx * n + k];
            L[ix * n + jx] = t / ljj;
        }
    }
    return 0;
}
/* log N(x; mu, L L') given lower factor L */
static double mvn_logpdf(const double *x, const double *mu, const double *L, int n) {
    double y[ORC_MAXD], quad = 0.0, logdet = 0.0;
    for (int ix = 0; ix < n; ++ix) {
        double t = x[ix] - mu[ix];
        for (int k = 0; k < ix; ++k) t -= L[ix * n + k] * y[k];
        y[ix] = t / L[ix * n + ix];
        quad += y[ix] * y[ix];
        logdet += log(L[ix * n + ix]);
    }
    logdet *= 2.0;
    return -((double)n * LOG2PI + logdet + quad) / 2.0;
}

/* ------------------------------------------------------------------ helpers.jl */
/* ref: src/helpers.jl:173-181 compute_ESS */
double orc_compute_ess(const double *loglh, const double *w, const double *old_loglh, int64_t n, double phi_n,
                       double phi_n1) {
    double s = 0.0;
    double *nw = (double *)malloc(sizeof(double) * n);
    for (int64_t i = 0; i < n; ++i) {
        double old = old_loglh ? old_loglh[i] : 0.0;
        double inc = exp((phi_n1 - phi_n) * old + (phi_n - phi_n1) * loglh[i]);
        nw[i] = w[i] * inc;
        s += nw[i];
    }
    double s2 = 0.0, N = (double)n;
    for (int64_t i = 0; i < n; ++i) {
        double v = N * nw[i] / s;
        s2 += v * v;
    }
    free(nw);
    return N * N / s2;
}

/* compute_ESS with the two sums reduced over OpenMP threads (CPU-baseline variant 2 only): ESS = (Σv)²/Σv², the same quantity
   as N²/Σ(N v/Σv)² without the temporary */
static int g_ess_threads = 1;
static double compute_ess_omp(const double *loglh, const double *w, const double *old_loglh, int64_t n, double phi_n,
                              double phi_n1) {
    double s = 0.0, s2 = 0.0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_ess_threads) reduction(+ : s, s2)
#endif
    for (int64_t i = 0; i < n; ++i) {
        double old = old_loglh ? old_loglh[i] : 0.0;
        double v = w[i] * exp((phi_n1 - phi_n) * old + (phi_n - phi_n1) * loglh[i]);
        s += v;
        s2 += v * v;
    }
    return s * s / s2;
}

/* CPU-baseline variant 2 only: the adaptive-ϕ solve on a PERSISTENT team.  ESS(ϕ_k) - ESS_bar for K candidates per pass over the
   cloud; every thread accumulates the 2 K sums of its chunk, thread 0 combines them in thread order and takes the decision while
   the others spin on a sequence number - one parallel region per solve.  With a passive wait policy (bench.py sets it: spinning
   workers slow the serial stretches of the run) every parallel region costs a wake-up of the whole team, and the sixty-odd regions
   per stage of a bisection to adjacent floats were what made this variant no faster than the serial one. */
#define ORC_KSEC 4
typedef struct {
    const double *loglh, *w, *old;
    int64_t n;
    double phi_n1, ess_bar;
    double phis[ORC_KSEC];
    int K, nt;
    double *part;                       /* [nt][2][ORC_KSEC] */
    _Atomic int seq, done, quit;
} ess_team;
static void team_chunk(ess_team *tm, int t) {
    const int64_t beg = tm->n * t / tm->nt, end = tm->n * (t + 1) / tm->nt;
    const int K = tm->K;
    double s[ORC_KSEC], s2[ORC_KSEC];
    for (int k = 0; k < K; ++k) s[k] = s2[k] = 0.0;
    for (int64_t i = beg; i < end; ++i) {
        const double old = tm->old ? tm->old[i] : 0.0, l = tm->loglh[i], wi = tm->w[i];
        for (int k = 0; k < K; ++k) {
            const double v = wi * exp((tm->phi_n1 - tm->phis[k]) * old + (tm->phis[k] - tm->phi_n1) * l);
            s[k] += v;
            s2[k] += v * v;
        }
    }
    for (int k = 0; k < K; ++k) { tm->part[((size_t)t * 2) * ORC_KSEC + k] = s[k]; tm->part[((size_t)t * 2 + 1) * ORC_KSEC + k] = s2[k]; }
}
/* thread 0: g_out[k] = ESS(phis[k]) - ESS_bar */
static void ess_candidates_team(ess_team *tm, const double *phis, int K, double *g_out) {
    for (int k = 0; k < K; ++k) tm->phis[k] = phis[k];
    tm->K = K;
    atomic_store(&tm->done, 0);
    atomic_fetch_add(&tm->seq, 1);                          /* release the workers */
    team_chunk(tm, 0);
    while (atomic_load(&tm->done) < tm->nt - 1) { }
    for (int k = 0; k < K; ++k) {
        double s = 0.0, s2 = 0.0;
        for (int t = 0; t < tm->nt; ++t) { s += tm->part[((size_t)t * 2) * ORC_KSEC + k]; s2 += tm->part[((size_t)t * 2 + 1) * ORC_KSEC + k]; }
        g_out[k] = s * s / s2 - tm->ess_bar;
    }
}
static void team_worker(ess_team *tm, int t) {
    int seen = 0;
    for (;;) {
        int sq;
        while ((sq = atomic_load(&tm->seq)) == seen) { if (atomic_load(&tm->quit)) return; }
        seen = sq;
        team_chunk(tm, t);
        atomic_fetch_add(&tm->done, 1);
    }
}
static double bit_at(double a, double b, int k, int K) {      /* the bit pattern a + (b - a) k / K, a <= b, both >= 0 */
    uint64_t ia, ib;
    memcpy(&ia, &a, 8);
    memcpy(&ib, &b, 8);
    const uint64_t span = ib - ia;
    uint64_t im = ia + (uint64_t)(((unsigned __int128)span * (unsigned)k) / (unsigned)K);
    double m;
    memcpy(&m, &im, 8);
    return m;
}
/* solve_adaptive_ϕ (helpers.jl:9-56) by K-section instead of bisection: the schedule walk takes K schedule points per pass, the root
   search K - 1 interior points of the bracket per pass (over the bit pattern, like Roots' bisection) - the same root to adjacent
   floats in ~11 passes instead of ~60 (for a monotone ESS; the ESS values themselves differ from the serial sums in the last bits). */
static int ksection_master(ess_team *tm, const double *sched, int32_t n_phi, int32_t *j, double *phi_prop, double phi_n1, double *phi_n, int *evals) {
    double g[ORC_KSEC], ph[ORC_KSEC];
    /* the walk: while g(ϕ_prop) >= 0 and j <= n_Φ: ϕ_prop = schedule[j]; j += 1 (helpers.jl:29-32), K schedule points per pass */
    double g_prop;
    for (;;) {
        int K = 0;
        ph[K++] = *phi_prop;
        for (int q = 0; K < ORC_KSEC && *j + q <= n_phi; ++q) ph[K++] = sched[*j - 1 + q];
        ess_candidates_team(tm, ph, K, g);
        *evals += K;
        int k = 0;
        while (k < K - 1 && g[k] >= 0.0) { *phi_prop = ph[k + 1]; *j += 1; ++k; }
        g_prop = g[k];
        if (!(g_prop >= 0.0 && *j <= n_phi && k == K - 1 && K > 1)) break;      /* walked off this batch with the condition still true: next batch */
    }
    if (*phi_prop != 1.0 || g_prop < 0.0) {
        double a = phi_n1, b = *phi_prop, fa, fb = g_prop;
        ph[0] = a;
        ess_candidates_team(tm, ph, 1, g);
        *evals += 1;
        fa = g[0];
        if (fa == 0.0) { *phi_n = a; return 0; }
        if (fb == 0.0) { *phi_n = b; return 0; }
        if ((fa > 0) == (fb > 0) || isnan(fa) || isnan(fb)) return fail("solve_adaptive_phi: bracket does not change sign");
        for (;;) {
            int K = 0;
            for (int k = 1; k <= ORC_KSEC; ++k) {             /* K interior points of the bracket, over the bit pattern like Roots' bisection */
                const double x = bit_at(a, b, k, ORC_KSEC + 1);
                if (x != a && x != b && (K == 0 || x != ph[K - 1])) ph[K++] = x;
            }
            if (K == 0) break;                                   /* adjacent floats */
            ess_candidates_team(tm, ph, K, g);
            *evals += K;
            int done = 0;
            for (int k = 0; k < K; ++k) {
                if (g[k] == 0.0 || isnan(g[k])) { *phi_n = ph[k]; done = 1; break; }
                if ((g[k] > 0) != (fa > 0)) { b = ph[k]; fb = g[k]; break; }
                a = ph[k]; fa = g[k];
            }
            if (done) return 0;
        }
        *phi_n = (fabs(fa) <= fabs(fb)) ? a : b;
    } else *phi_n = 1.0;
    return 0;
}
/* solve_adaptive_ϕ (helpers.jl:9-56) by K-section on a persistent team: the same root to adjacent floats (for a monotone ESS; the ESS
   values themselves differ from the serial sums in the last bits). */
static int solve_adaptive_phi_ksection(const double *loglh, const double *w, const double *old, int64_t n, double ess_bar, const double *sched,
                                       int32_t n_phi, int32_t *j, double *phi_prop, double phi_n1, double *phi_n, int *evals) {
    ess_team tm;
    memset(&tm, 0, sizeof(tm));
    tm.loglh = loglh; tm.w = w; tm.old = old; tm.n = n; tm.phi_n1 = phi_n1; tm.ess_bar = ess_bar;
    int nt = g_ess_threads;
    if ((int64_t)nt * 1024 > n) nt = (int)(n / 1024) > 0 ? (int)(n / 1024) : 1;
    tm.part = (double *)calloc((size_t)nt * 2 * ORC_KSEC, sizeof(double));
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
    {
#pragma omp single
        tm.nt = omp_get_num_threads();                         /* (implicit barrier: every thread sees the team size) */
        const int t = omp_get_thread_num();
        if (t == 0) {
            rc = ksection_master(&tm, sched, n_phi, j, phi_prop, phi_n1, phi_n, evals);
            atomic_store(&tm.quit, 1);
        } else team_worker(&tm, t);
    }
#else
    tm.nt = 1;
    rc = ksection_master(&tm, sched, n_phi, j, phi_prop, phi_n1, phi_n, evals);
#endif
    free(tm.part);
    return rc;
}

static double bit_middle(double a, double b) { /* Roots.jl exact bisection midpoint over the bit pattern, a,b >= 0 */
    uint64_t ia, ib;
    memcpy(&ia, &a, 8);
    memcpy(&ib, &b, 8);
    uint64_t im = ia + ((ib - ia) >> 1);
    double m;
    memcpy(&m, &im, 8);
    return m;
}

/* ref: src/helpers.jl:9-56 solve_adaptive_ϕ.  j is the reference's 1-based schedule index. */
int orc_solve_adaptive_phi(const double *particles, int64_t n, int32_t R, double ess_prev, const double *sched,
                           int32_t n_phi, int32_t *j, double *phi_prop, double phi_n1, double target,
                           int32_t *resampled_last, double *phi_n, int32_t *n_evals) {
    const double *loglh = particles + (int64_t)(R - 5) * n, *old = particles + (int64_t)(R - 3) * n,
                 *w = particles + (int64_t)(R - 1) * n;
    double ess_bar;
    int evals = 0;
    if (*resampled_last) { ess_bar = target * (double)n; *resampled_last = 0; }   /* helpers.jl:14-20 */
    else ess_bar = target * ess_prev;
    if (g_ess_threads > 1) {                                       /* CPU-baseline variant 2: K candidates per pass */
        int rc = solve_adaptive_phi_ksection(loglh, w, old, n, ess_bar, sched, n_phi, j, phi_prop, phi_n1, phi_n, &evals);
        if (n_evals) *n_evals = evals;
        return rc;
    }
#define G(phi) (evals++, (g_ess_threads > 1 ? compute_ess_omp(loglh, w, old, n, (phi), phi_n1) : orc_compute_ess(loglh, w, old, n, (phi), phi_n1)) - ess_bar)
    while (G(*phi_prop) >= 0.0 && *j <= n_phi) { *phi_prop = sched[*j - 1]; *j += 1; } /* helpers.jl:29-32 */
    if (*phi_prop != 1.0 || G(*phi_prop) < 0.0) {                                 /* helpers.jl:48-50 */
        /* Roots.fzero(g, [ϕ_n1, ϕ_prop], xtol = 0.): bisection over the bit pattern to adjacent floats */
        double a = phi_n1, b = *phi_prop, fa = G(a), fb = G(b), root;
        if (fa == 0.0) root = a;
        else if (fb == 0.0) root = b;
        else if ((fa > 0) == (fb > 0) || isnan(fa) || isnan(fb)) return fail("solve_adaptive_phi: bracket does not change sign");
        else {
            int done = 0;
            root = a;
            for (;;) {
                double mm = bit_middle(a, b);
                if (mm == a || mm == b) break;
                double fm = G(mm);
                if (fm == 0.0 || isnan(fm)) { root = mm; done = 1; break; }
                if ((fm > 0) != (fa > 0)) { b = mm; fb = fm; } else { a = mm; fa = fm; }
            }
            if (!done) root = (fabs(fa) <= fabs(fb)) ? a : b;
        }
        *phi_n = root;
    } else {
        *phi_n = 1.0;                                                              /* helpers.jl:51-53 */
    }
#undef G
    if (n_evals) *n_evals = evals;
    return 0;
}

/* ref: src/helpers.jl:128-164 compute_proposal_densities (DegenerateMvNormal logpdf restated as the
   full-rank MVN log-density; reproduced against proposal_densities_in.jld2 to 1e-15) */
static void proposal_densities_L(const double *para_draw, const double *para_subset, const double *mu,
                                 const double *Sigma, const double *L /* chol(c^2 Sigma) */, int db, double alpha,
                                 double *q0o, double *q1o) {
    double q0 = alpha * exp(mvn_logpdf(para_subset, para_draw, L, db));
    double q1 = alpha * exp(mvn_logpdf(para_draw, para_subset, L, db));
    double ind_pdf = 1.0;
    for (int i = 0; i < db; ++i) {
        double sii = sqrt(Sigma[i * db + i]);            /* NOT scaled by c^2: quirk Q1, helpers.jl:146-148 */
        double z = (para_subset[i] - para_draw[i]) / sii;
        ind_pdf = ind_pdf / (sii * sqrt(2.0 * M_PI)) * exp(-0.5 * z * z);
    }
    q0 += (1.0 - alpha) / 2.0 * ind_pdf;
    q1 += (1.0 - alpha) / 2.0 * ind_pdf;
    q0 += (1.0 - alpha) / 2.0 * exp(mvn_logpdf(para_subset, mu, L, db));
    q1 += (1.0 - alpha) / 2.0 * exp(mvn_logpdf(para_draw, mu, L, db));
    q0 = log(q0);
    q1 = log(q1);
    if (q0 == INFINITY && q1 == INFINITY) q0 = 0.0;
    *q0o = q0;
    *q1o = q1;
}
static int scaled_chol(const double *Sigma, int db, double c, double *L) {
    double *S = (double *)malloc(sizeof(double) * db * db);
    for (int i = 0; i < db * db; ++i) S[i] = c * c * Sigma[i];        /* c^2 * d_prop.Σ, helpers.jl:90 */
    int rc = chol_lower(S, db, L);
    free(S);
    return rc;
}
void orc_proposal_densities(const double *para_draw, const double *para_subset, const double *mu,
                            const double *Sigma, int32_t db, double c, double alpha, double *q0, double *q1) {
    double *L = (double *)malloc(sizeof(double) * db * db);
    if (scaled_chol(Sigma, db, c, L) != 0) { *q0 = *q1 = NAN; free(L); return; }
    proposal_densities_L(para_draw, para_subset, mu, Sigma, L, db, alpha, q0, q1);
    free(L);
}

/* ref: src/helpers.jl:87-100 mvnormal_mixture_draw.  Philox: tag(P_MUT,t,0).ua picks the component,
   tag(P_MUT,t,1+i/2) gives the Box-Muller pair for elements i,i+1. */
static void mixture_draw_L(const double *theta_old, const double *mu, const double *Sigma, const double *L, int db,
                           double c, double alpha, uint64_t seed, uint64_t pid, uint32_t stage, uint32_t t,
                           double *theta_new) {
    double uc, unext, z[ORC_MAXD];
    orc_uniform_pair(seed, pid, stage, TAG(P_MUT, t, 0), &uc, &unext);
    for (int i = 0; i < db; i += 2) {
        double z0, z1;
        normal_pair(seed, pid, stage, TAG(P_MUT, t, 1 + i / 2), &z0, &z1);
        z[i] = z0;
        if (i + 1 < db) z[i + 1] = z1;
    }
    int comp = (uc < alpha) ? 0 : (uc < alpha + (1.0 - alpha) / 2.0 ? 1 : 2);
    if (comp == 1) {
        for (int i = 0; i < db; ++i) theta_new[i] = theta_old[i] + sqrt(c * c * Sigma[i * db + i]) * z[i];
    } else {
        const double *center = (comp == 0) ? theta_old : mu;
        for (int i = 0; i < db; ++i) {
            double s = 0.0;
            for (int k = 0; k <= i; ++k) s += L[i * db + k] * z[k];
            theta_new[i] = center[i] + s;
        }
    }
}
int orc_mixture_draw(const double *theta_old, const double *mu, const double *Sigma, int32_t db, double c,
                     double alpha, uint64_t seed, uint64_t pid, uint32_t stage, uint32_t t, double *theta_new) {
    double *L = (double *)malloc(sizeof(double) * db * db);
    if (scaled_chol(Sigma, db, c, L) != 0) { free(L); return fail("mixture_draw: covariance not positive definite"); }
    mixture_draw_L(theta_old, mu, Sigma, L, db, c, alpha, seed, pid, stage, t, theta_new);
    free(L);
    return 0;
}

/* ref: src/helpers.jl:215-231 generate_free_blocks + :244-260 generate_all_blocks.
   shuffle(1:n_free) = Fisher-Yates on Philox tag(P_BLK, i, 0).ua.  Indices are 0-based here. */
void orc_generate_blocks(int32_t n_free, int32_t n_blocks, const int32_t *free_inds, uint64_t seed, uint32_t stage,
                         int32_t *blocks_free, int32_t *blocks_all, int32_t *block_ptr) {
    for (int i = 0; i < n_free; ++i) blocks_free[i] = i;
    for (int i = n_free - 1; i >= 1; --i) {
        double ua, ub;
        orc_uniform_pair(seed, 0, stage, TAG(P_BLK, i, 0), &ua, &ub);
        int jx = (int)(ua * (double)(i + 1));
        if (jx > i) jx = i;
        int tmp = blocks_free[i]; blocks_free[i] = blocks_free[jx]; blocks_free[jx] = tmp;
    }
    int sub = (n_free + n_blocks - 1) / n_blocks;      /* cld */
    for (int b = 0; b < n_blocks; ++b) block_ptr[b] = b * sub;
    block_ptr[n_blocks] = n_free;                      /* last block takes the remainder (shorter) */
    for (int i = 0; i < n_free; ++i) blocks_all[i] = free_inds[blocks_free[i]];
}

/* ------------------------------------------------------------------ particle.jl / smc_main.jl */
/* ref: src/smc_main.jl:401-420 incremental weights; src/particle.jl:250-256 update_weights!;
   :362-366 normalize_weights!; smc_main.jl:427 ESS */
void orc_correct(double *particles, int64_t n, int32_t R, double phi_n, double phi_n1, double pw, double logp_old,
                 double *inc_w, double *norm_w, double *ess, double *sum_unnorm) {
    const double *loglh = particles + (int64_t)(R - 5) * n, *old = particles + (int64_t)(R - 3) * n;
    double *w = particles + (int64_t)(R - 1) * n;
    double N = (double)n, s = 0.0, s2 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double inc;
        if (pw == 0.0) inc = exp((phi_n1 - phi_n) * old[i] + (phi_n - phi_n1) * loglh[i]);
        else if (pw == 1.0) inc = exp((phi_n - phi_n1) * loglh[i]);
        else inc = exp((phi_n1 - phi_n) * log(exp(old[i] - logp_old + log(1.0 - pw)) + pw) + (phi_n - phi_n1) * loglh[i]);
        inc_w[i] = inc;
        w[i] *= inc;
    }
    for (int64_t i = 0; i < n; ++i) s += w[i];
    for (int64_t i = 0; i < n; ++i) { w[i] *= N; w[i] /= s; norm_w[i] = w[i]; s2 += w[i] * w[i]; }
    *ess = N * N / s2;
    *sum_unnorm = s;
}

/* ref: src/resample.jl:23-72.  Output indices 0-based.  Fall-through (reference returns 0 / nothing,
   reachable only through round-off) is clamped to the last index - the one documented deviation. */
void orc_resample_with_offsets(const double *weights, int64_t nw, int64_t n_parts, int32_t method,
                               const double *offsets, int64_t *idx) {
    double *cw = (double *)malloc(sizeof(double) * nw), s = 0.0, run = 0.0;
    for (int64_t i = 0; i < nw; ++i) s += weights[i];
    for (int64_t i = 0; i < nw; ++i) { run += weights[i] / s; cw[i] = run; }  /* cumsum(weights ./ sum(weights)) */
    if (method == ORC_RESAMPLE_MULTINOMIAL) {
        for (int64_t i = 0; i < n_parts; ++i) {                                  /* resample.jl:38-41 */
            int64_t f = -1;
            for (int64_t jx = 0; jx < nw; ++jx) if (offsets[i] < cw[jx]) { f = jx; break; }
            idx[i] = f < 0 ? nw - 1 : f;
        }
    } else {
        double offset = offsets[0];
        int64_t start = 0, lim = n_parts < nw ? n_parts : nw;                    /* range start_ind:n_parts, quirk Q5 */
        for (int64_t i = 0; i < n_parts; ++i) {
            double thr = ((double)i + offset) / (double)n_parts;                /* (i - 1 + offset)/n_parts, 1-based i */
            int64_t f = -1;
            for (int64_t jx = start; jx < lim; ++jx) if (cw[jx] > thr) { f = jx; break; }
            if (f < 0) f = lim - 1;
            idx[i] = f;
            start = f;
        }
    }
    free(cw);
}
void orc_resample(const double *weights, int64_t nw, int64_t n_parts, int32_t method, uint64_t seed, uint32_t stage,
                  int64_t *idx) {
    if (method == ORC_RESAMPLE_MULTINOMIAL) {
        double *u = (double *)malloc(sizeof(double) * n_parts), ub;
        for (int64_t i = 0; i < n_parts; ++i) orc_uniform_pair(seed, (uint64_t)i, stage, TAG(P_RES, 0, 0), &u[i], &ub);
        orc_resample_with_offsets(weights, nw, n_parts, method, u, idx);
        free(u);
    } else {
        double u, ub;
        orc_uniform_pair(seed, 0, stage, TAG(P_RES, 0, 0), &u, &ub);
        orc_resample_with_offsets(weights, nw, n_parts, method, &u, idx);
    }
}

/* ref: src/particle.jl:481-483 weighted_mean */
void orc_weighted_mean(const double *particles, int64_t n, int32_t R, double *mean) {
    const double *w = particles + (int64_t)(R - 1) * n;
    double sw = 0.0;
    for (int64_t i = 0; i < n; ++i) sw += w[i];
    for (int k = 0; k < R - 5; ++k) {
        const double *x = particles + (int64_t)k * n;
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += x[i] * w[i];
        mean[k] = s / sw;
    }
}
/* ref: src/particle.jl:526-529 weighted_cov = StatsBase.cov(X, Weights(W/ΣW), corrected=false):
   weighted mean, then Σ w (x-m)(x-m)' / Σw.  Output row-major d x d. */
void orc_weighted_cov(const double *particles, int64_t n, int32_t R, double *cov) {
    int d = R - 5;
    const double *w = particles + (int64_t)(R - 1) * n;
    double sw = 0.0, swn = 0.0;
    double *wn = (double *)malloc(sizeof(double) * n), *m = (double *)malloc(sizeof(double) * d);
    for (int64_t i = 0; i < n; ++i) sw += w[i];
    for (int64_t i = 0; i < n; ++i) { wn[i] = w[i] / sw; swn += wn[i]; }
    for (int k = 0; k < d; ++k) {
        const double *x = particles + (int64_t)k * n;
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += wn[i] * x[i];
        m[k] = s / swn;
    }
    /* (CPU-baseline variant 2: the d (d + 1) / 2 entries over the team, each still summed serially - the same bits as the serial loop) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_ess_threads) if (g_ess_threads > 1)
#endif
    for (int ab = 0; ab < d * d; ++ab) {
        const int a = ab / d, b = ab % d;
        if (b < a) continue;
        const double *xa = particles + (int64_t)a * n, *xb = particles + (int64_t)b * n;
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += wn[i] * (xa[i] - m[a]) * (xb[i] - m[b]);
        cov[a * d + b] = cov[b * d + a] = s * (1.0 / swn);
    }
    free(wn);
    free(m);
}
/* ref: src/smc_main.jl:453-455 */
double orc_update_c(double c, double accept, double target) {
    return c * (0.95 + 0.10 * exp(16.0 * (accept - target)) / (1.0 + exp(16.0 * (accept - target))));
}

/* ------------------------------------------------------------------ model: priors, bounds, likelihoods */
/* ModelConstructors.update! bounds check (closed interval) - dependency absent, behaviour per docs */
int orc_in_bounds(const orc_model *m, const double *theta) {
    for (int k = 0; k < m->n_para; ++k)
        if (!(m->lo[k] <= theta[k] && theta[k] <= m->hi[k])) return 0;
    return 1;
}
static double prior_logpdf(int fam, double a, double b, double x) {
    switch (fam) {
    case ORC_PRIOR_NORMAL: { double z = (x - a) / b; return -(z * z + LOG2PI) / 2.0 - log(b); }
    case ORC_PRIOR_UNIFORM: return (a <= x && x <= b) ? -log(b - a) : -INFINITY;
    case ORC_PRIOR_GAMMA: return x < 0 ? -INFINITY : -lgamma(a) - a * log(b) + (a - 1.0) * log(x) - x / b;
    case ORC_PRIOR_BETA:
        return (x < 0 || x > 1) ? -INFINITY
                                : (a - 1.0) * log(x) + (b - 1.0) * log1p(-x) - (lgamma(a) + lgamma(b) - lgamma(a + b));
    case ORC_PRIOR_INVGAMMA: return x <= 0 ? -INFINITY : a * log(b) - lgamma(a) - (a + 1.0) * log(x) - b / x;
    case ORC_PRIOR_ROOTINVGAMMA:
        return x <= 0 ? -INFINITY
                      : log(2.0) - lgamma(a / 2.0) + (a / 2.0) * log(a * b * b / 2.0) - ((a + 1.0) / 2.0) * log(x * x) -
                            a * b * b / (2.0 * x * x);
    default: return NAN;
    }
}
/* ModelConstructors.prior(parameters): Σ logpdf over free parameters (pinned for Normal/Uniform by the
   400+400 stored logpriors of test/reference/initial_draw_out_*, initialize_likelihood_out_*) */
double orc_logprior(const orc_model *m, const double *theta) {
    double s = 0.0;
    for (int k = 0; k < m->n_para; ++k)
        if (!m->fixed[k]) s += prior_logpdf(m->prior_family[k], m->prior_a[k], m->prior_b[k], theta[k]);
    return s;
}

/* Linear-Gaussian state-space likelihood by the Kalman filter - the same statement order as the device code (csrc/model.hpp). */
static double orc_kalman_lgss(const orc_lik *l, const double *th) {
#define TH(k) th[k]
#define AUX l->aux
#define PAR0 l->par[0]
#define YDAT l->data
#define NT l->cols
#define NEGINF (-INFINITY)
    /* LGSS_KALMAN (build-defined, no reference source).  n_s = 8 states, n_y = 3 observables, n_r = 3 shocks,
       d = 13 parameters: th[0..7] = ρ (diagonal of the transition), th[8..10] = shock std σ, th[11] = measurement std σ_e,
       th[12] = measurement mean μ.  x_t = Tm x_{t-1} + Rm ε_t, ε ~ N(0, diag σ²); y_t = μ + Z x_t + u_t, u ~ N(0, σ_e² I);
       Tm = diag(ρ) + κ C.  aux = [C (8x8) | Rm (8x3) | Z (3x8)] row-major, par[0] = κ, data = y (3 x T, column-major).
       x_0 = 0, P_0 = I.  Innovations form with a 3x3 Cholesky of F_t; returns -Inf when F_t is not positive definite. */
    const double *Cm = AUX, *Rm = AUX + 64, *Zm = AUX + 88;
    const double kappa = PAR0, mu = TH(12), se2 = TH(11) * TH(11);
    double Tm[64], RQR[64], P[64], TP[64], x[8], xp[8], PZ[24], G[24];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            Tm[i * 8 + j] = (i == j ? TH(i) : 0.0) + kappa * Cm[i * 8 + j];
            double s = 0.0;
            for (int m = 0; m < 3; ++m) s += Rm[i * 3 + m] * (TH(8 + m) * TH(8 + m)) * Rm[j * 3 + m];
            RQR[i * 8 + j] = s;
            P[i * 8 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int i = 0; i < 8; ++i) x[i] = 0.0;
    double ll = 0.0;
    for (long long t = 0; t < NT; ++t) {
        for (int i = 0; i < 8; ++i) {
            double s = 0.0;
            for (int j = 0; j < 8; ++j) s += Tm[i * 8 + j] * x[j];
            xp[i] = s;
        }
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) {
                double s = 0.0;
                for (int k = 0; k < 8; ++k) s += Tm[i * 8 + k] * P[k * 8 + j];
                TP[i * 8 + j] = s;
            }
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) {
                double s = RQR[i * 8 + j];
                for (int k = 0; k < 8; ++k) s += TP[i * 8 + k] * Tm[j * 8 + k];
                P[i * 8 + j] = s;                                   /* P_{t|t-1} */
            }
        double v[3], F[9];
        for (int a = 0; a < 3; ++a) {
            double s = 0.0;
            for (int j = 0; j < 8; ++j) s += Zm[a * 8 + j] * xp[j];
            v[a] = YDAT[a + 3 * t] - mu - s;
        }
        for (int i = 0; i < 8; ++i)
            for (int a = 0; a < 3; ++a) {
                double s = 0.0;
                for (int j = 0; j < 8; ++j) s += P[i * 8 + j] * Zm[a * 8 + j];
                PZ[i * 3 + a] = s;
            }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = (a == b) ? se2 : 0.0;
                for (int i = 0; i < 8; ++i) s += Zm[a * 8 + i] * PZ[i * 3 + b];
                F[a * 3 + b] = s;
            }
        if (!(F[0] > 0.0)) return NEGINF;
        const double l00 = sqrt(F[0]), l10 = F[3] / l00, l20 = F[6] / l00;
        const double p11 = F[4] - l10 * l10;
        if (!(p11 > 0.0)) return NEGINF;
        const double l11 = sqrt(p11), l21 = (F[7] - l20 * l10) / l11;
        const double p22 = F[8] - l20 * l20 - l21 * l21;
        if (!(p22 > 0.0)) return NEGINF;
        const double l22 = sqrt(p22);
        const double w0 = v[0] / l00, w1 = (v[1] - l10 * w0) / l11, w2 = (v[2] - l20 * w0 - l21 * w1) / l22;
        ll += -1.5 * log(2.0 * M_PI) - (log(l00) + log(l11) + log(l22)) - 0.5 * (w0 * w0 + w1 * w1 + w2 * w2);
        const double u2 = w2 / l22, u1 = (w1 - l21 * u2) / l11, u0 = (w0 - l10 * u1 - l20 * u2) / l00;
        for (int i = 0; i < 8; ++i) {
            x[i] = xp[i] + (PZ[i * 3 + 0] * u0 + PZ[i * 3 + 1] * u1 + PZ[i * 3 + 2] * u2);
            const double g0 = PZ[i * 3 + 0] / l00, g1 = (PZ[i * 3 + 1] - l10 * g0) / l11;
            G[i * 3 + 0] = g0; G[i * 3 + 1] = g1; G[i * 3 + 2] = (PZ[i * 3 + 2] - l20 * g0 - l21 * g1) / l22;
        }
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j)
                P[i * 8 + j] -= G[i * 3 + 0] * G[j * 3 + 0] + G[i * 3 + 1] * G[j * 3 + 1] + G[i * 3 + 2] * G[j * 3 + 2];
    }
    return ll;
#undef TH
#undef AUX
#undef PAR0
#undef YDAT
#undef NT
#undef NEGINF
}

double orc_loglik(const orc_lik *l, const double *th, int32_t d) {
    switch (l->family) {
    case ORC_LIK_GAUSS_ISO: { /* ℓ = -(d/2) log(2π σ²) - Σ(θ_j - m_j)²/(2σ²) */
        double sig = l->par[0], acc = 0.0;
        for (int k = 0; k < d; ++k) { double e = th[k] - l->data[k]; acc += e * e; }
        return -0.5 * (double)d * log(2.0 * M_PI * sig * sig) - acc / (2.0 * sig * sig);
    }
    case ORC_LIK_LINREG: { /* ref: examples/regression_model/estimate_regression.jl:46-53; data = [y X] (n x 2) */
        int64_t n = l->rows;
        const double *y = l->data, *X = l->data + n;
        double s2 = l->par[0], Nn = (double)n, dot = 0.0;
        double term1 = -(Nn / 2.0) * log(2.0 * M_PI) - (Nn / 2.0) * log(s2);
        for (int64_t t = 0; t < n; ++t) { double e = y[t] - th[0] - th[1] * X[t]; dot += e * e; }
        return term1 - (1.0 / (2.0 * s2)) * dot;
    }
    case ORC_LIK_LINMODEL3: { /* ref: test/modelsetup.jl:119-138 loglik_fn; data 3 x T, aux = X 3 x (>=T) */
        int64_t T = l->cols;
        double a[3], b[3], inv[3], det = 1.0;
        for (int i = 0; i < 3; ++i) {
            a[i] = th[3 * i]; b[i] = th[3 * i + 1];
            double v = th[3 * i + 2] * th[3 * i + 2];
            if (v == 0.0) return -INFINITY;            /* inv(Σ) SingularException -> caught -> -Inf */
            det *= v; inv[i] = 1.0 / v;
        }
        double term1 = -3.0 / 2.0 * log(2.0 * M_PI) - 1.0 / 2.0 * log(det), lp = 0.0;
        for (int64_t t = 0; t < T; ++t) {
            double q = 0.0;
            for (int i = 0; i < 3; ++i) {
                double e = l->data[i + 3 * t] - a[i] - b[i] * l->aux[i + l->aux_rows * t];
                q += e * (inv[i] * e);
            }
            lp += term1 - 1.0 / 2.0 * q;
        }
        return lp;
    }
    case ORC_LIK_LGSS_KALMAN: return (d == 13 && l->rows == 3 && l->aux_rows * l->aux_cols >= 112) ? orc_kalman_lgss(l, th) : NAN;
    case ORC_LIK_CAPM_LITERAL: { /* ref: examples/capm_model/estimate_capm.jl:52-70 AS WRITTEN (quirk Q12):
                                    β_i := p[3i-2] (= α_i) and the full 3xT quadratic form inside the t loop */
        int64_t T = l->cols;
        double a[3], inv[3], det = 1.0;
        for (int i = 0; i < 3; ++i) {
            a[i] = th[3 * i];
            double v = th[3 * i + 2] * th[3 * i + 2];
            if (v == 0.0) return -INFINITY;
            det *= v; inv[i] = 1.0 / v;
        }
        double term1 = -3.0 / 2.0 * log(2.0 * M_PI) - 1.0 / 2.0 * log(det), S = 0.0, lp = 0.0;
        for (int64_t t = 0; t < T; ++t)
            for (int i = 0; i < 3; ++i) {
                double e = l->data[i + 3 * t] - a[i] - a[i] * l->aux[l->aux_rows * t];
                S += e * (inv[i] * e);
            }
        for (int64_t t = 0; t < T; ++t) lp += term1 - 1.0 / 2.0 * S;
        return lp;
    }
    default: return NAN;
    }
}

/* ------------------------------------------------------------------ mutation.jl */
typedef struct { int db; double *L; double *Sig; double *mu; int ok; } blk_factor;

/* ref: src/mutation.jl:56-138.  p is one cloud row (length R) addressed with `stride`. */
static int mutation_core(const orc_model *m, double *p, int64_t stride, const blk_factor *bf, int32_t n_free,
                         const int32_t *blocks_all, const int32_t *block_ptr, int32_t n_blocks, double phi_n,
                         double c, double alpha, int32_t n_mh_steps, uint64_t seed, uint64_t pid, uint32_t stage) {
    int d = m->n_para;
    double para[ORC_MAXD], para_new[ORC_MAXD], sub[ORC_MAXD], draw[ORC_MAXD];
    for (int k = 0; k < d; ++k) para[k] = p[(int64_t)k * stride];
    double like = p[(int64_t)d * stride], logprior = p[(int64_t)(d + 1) * stride],
           like_prev = p[(int64_t)(d + 2) * stride], accept = 0.0, step_prob, dummy;
    orc_uniform_pair(seed, pid, stage, TAG(P_MUT, 0xFFFFF, 0), &step_prob, &dummy);       /* mutation.jl:66 */
    for (int step = 0; step < n_mh_steps; ++step)
        for (int b = 0; b < n_blocks; ++b) {
            const int32_t *ba = blocks_all + block_ptr[b];
            const blk_factor *f = &bf[b];
            int db = f->db;
            uint32_t t = (uint32_t)(step * n_blocks + b);
            if (!f->ok) return fail("mutation: block covariance not positive definite (PosDefException)");
            for (int i = 0; i < db; ++i) sub[i] = para[ba[i]];
            mixture_draw_L(sub, f->mu, f->Sig, f->L, db, c, alpha, seed, pid, stage, t, draw);
            double q0, q1;
            proposal_densities_L(draw, sub, f->mu, f->Sig, f->L, db, alpha, &q0, &q1);
            memcpy(para_new, para, sizeof(double) * d);
            for (int i = 0; i < db; ++i) para_new[ba[i]] = draw[i];
            double prior_new = -INFINITY, like_new = -INFINITY, like_old_data = -INFINITY;
            if (orc_in_bounds(m, para_new)) {                                             /* update! :93 */
                prior_new = orc_logprior(m, para_new);                                     /* :95 */
                like_new = orc_loglik(&m->lik, para_new, d);                               /* :96 */
                if (like_new == -INFINITY) prior_new = like_old_data = -INFINITY;          /* :102-104 */
                like_old_data = (m->old_lik.family == ORC_LIK_NONE) ? 0.0 : orc_loglik(&m->old_lik, para_new, d);
            }
            double eta = exp(phi_n * (like_new - like) + (1.0 - phi_n) * (like_old_data - like_prev) +
                             (prior_new - logprior) + (q0 - q1));                          /* :123-124 */
            if (step_prob < eta) {                                                         /* :126-132 */
                memcpy(para, para_new, sizeof(double) * d);
                like = like_new; logprior = prior_new; like_prev = like_old_data;
                accept += (double)db;
            }
            double ucomp;
            orc_uniform_pair(seed, pid, stage, TAG(P_MUT, t, 0), &ucomp, &step_prob);    /* :133 next step_prob */
        }
    for (int k = 0; k < d; ++k) p[(int64_t)k * stride] = para[k];                          /* update_mutation! */
    p[(int64_t)d * stride] = like;
    p[(int64_t)(d + 1) * stride] = logprior;
    p[(int64_t)(d + 2) * stride] = like_prev;
    p[(int64_t)(d + 3) * stride] = accept / (double)n_free;                               /* quirk Q2 */
    return 0;
}

static blk_factor *make_factors(const double *mu_free, const double *Sigma_free, int n_free,
                                const int32_t *blocks_free, const int32_t *block_ptr, int n_blocks, double c) {
    blk_factor *bf = (blk_factor *)calloc(n_blocks, sizeof(blk_factor));
    for (int b = 0; b < n_blocks; ++b) {
        int db = block_ptr[b + 1] - block_ptr[b];
        const int32_t *idx = blocks_free + block_ptr[b];
        bf[b].db = db;
        bf[b].L = (double *)malloc(sizeof(double) * db * db);
        bf[b].Sig = (double *)malloc(sizeof(double) * db * db);
        bf[b].mu = (double *)malloc(sizeof(double) * db);
        for (int i = 0; i < db; ++i) {
            bf[b].mu[i] = mu_free[idx[i]];
            for (int k = 0; k < db; ++k) bf[b].Sig[i * db + k] = Sigma_free[idx[i] * n_free + idx[k]];
        }
        bf[b].ok = (scaled_chol(bf[b].Sig, db, c, bf[b].L) == 0);
    }
    return bf;
}
static void free_factors(blk_factor *bf, int n_blocks) {
    for (int b = 0; b < n_blocks; ++b) { free(bf[b].L); free(bf[b].Sig); free(bf[b].mu); }
    free(bf);
}

int orc_mutation(const orc_model *m, double *p, int64_t stride, const double *mu_free, const double *Sigma_free,
                 int32_t n_free, const int32_t *blocks_free, const int32_t *blocks_all, const int32_t *block_ptr,
                 int32_t n_blocks, double phi_n, double phi_n1, double c, double alpha, int32_t n_mh_steps,
                 uint64_t seed, uint64_t pid, uint32_t stage) {
    (void)phi_n1;
    blk_factor *bf = make_factors(mu_free, Sigma_free, n_free, blocks_free, block_ptr, n_blocks, c);
    int rc = mutation_core(m, p, stride, bf, n_free, blocks_all, block_ptr, n_blocks, phi_n, c, alpha, n_mh_steps,
                           seed, pid, stage);
    free_factors(bf, n_blocks);
    return rc;
}

static int g_refactor_per_particle = 0;   /* CPU-baseline variant 1 (orc_run_config.variant) */
/* the reference's `[mutation_closure(cloud.particles[k,:], ...) for k=1:n_parts]` (smc_main.jl:472-481).
   The per-particle MvNormal/Cholesky of the reference (mutation.jl:81) is hoisted: same value for all k. */
int orc_mutate_cloud(const orc_model *m, double *particles, int64_t n, int64_t pid0, const double *mu_free,
                     const double *Sigma_free, int32_t n_free, const int32_t *blocks_free,
                     const int32_t *blocks_all, const int32_t *block_ptr, int32_t n_blocks, double phi_n,
                     double phi_n1, double c, double alpha, int32_t n_mh_steps, uint64_t seed, uint32_t stage,
                     int32_t n_threads) {
    (void)phi_n1;
    blk_factor *bf = make_factors(mu_free, Sigma_free, n_free, blocks_free, block_ptr, n_blocks, c);
    int rc = 0;
    const int refactor = g_refactor_per_particle;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(static) num_threads(n_threads) reduction(| : rc)
#endif
    for (int64_t i = 0; i < n; ++i) {
        if (refactor) {
            /* the reference's cost: MvNormal(θ̄_b, Σ_b) is built inside mutation(), i.e. once per particle (mutation.jl:81); the
               factors are the same bits as the hoisted ones, so the results do not change */
            blk_factor *bp = make_factors(mu_free, Sigma_free, n_free, blocks_free, block_ptr, n_blocks, c);
            rc |= mutation_core(m, particles + i, n, bp, n_free, blocks_all, block_ptr, n_blocks, phi_n, c, alpha,
                                n_mh_steps, seed, (uint64_t)(pid0 + i), stage) != 0;
            free_factors(bp, n_blocks);
        } else
        rc |= mutation_core(m, particles + i, n, bf, n_free, blocks_all, block_ptr, n_blocks, phi_n, c, alpha,
                            n_mh_steps, seed, (uint64_t)(pid0 + i), stage) != 0;
    }
    free_factors(bf, n_blocks);
    (void)n_threads;
    return rc ? fail("mutation: block covariance not positive definite (PosDefException)") : 0;
}

/* ------------------------------------------------------------------ initialization.jl */
/* ref: src/initialization.jl:153-186 initialize_likelihoods!: loglh -> old_loglh, then loglh / logpost(prior)
   re-evaluated on the current (new) data for every particle (draw_likelihood, :125-137).  The reference lets a
   ParamBoundsError escape here; out-of-bounds values are recorded as -Inf instead. */
void orc_initialize_likelihoods(const orc_model *m, double *particles, int64_t n) {
    int d = m->n_para;
    for (int64_t i = 0; i < n; ++i) {
        double th[ORC_MAXD];
        for (int k = 0; k < d; ++k) th[k] = particles[(int64_t)k * n + i];
        particles[(int64_t)(d + 2) * n + i] = particles[(int64_t)d * n + i];
        double ll = -INFINITY, lp = -INFINITY;
        if (orc_in_bounds(m, th)) { ll = orc_loglik(&m->lik, th, d); lp = orc_logprior(m, th); }
        particles[(int64_t)d * n + i] = ll;
        particles[(int64_t)(d + 1) * n + i] = lp;
    }
}

/* ref: src/initialization.jl:23-63 one_draw + :88-119 initial_draw!.  Philox: counter stage field = outer
   attempt, tag(P_INIT, redraw, k) per parameter.  Only Normal / Uniform priors can be sampled here. */
/* rand(::ParameterVector) for one parameter on the RNG contract: Normal / Uniform from one Philox call; Gamma, Beta,
   InverseGamma, RootInverseGamma through unit gammas by Marsaglia & Tsang (2000): iteration mm < 32 of unit gamma g reads its normal from
   tag(P_INIT, r | (32 g + mm) << 14, k) and its acceptance uniform u1 (u2: the shape < 1 boost) from tag(P_INIT, same, k | 64).
   Distributions.jl's own samplers differ (parity unpinned: the reference's streams are MersenneTwister's); the DISTRIBUTIONS are what
   ModelConstructors' prior(...) evaluates (prior_logpdf above). */
static double gamma_unit_draw(uint64_t seed, uint64_t pid, uint32_t attempt, uint32_t r, uint32_t k, uint32_t g, double shape) {
    int boost = shape < 1.0;
    double a = boost ? shape + 1.0 : shape;
    double dd = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
    for (uint32_t mm = 0; mm < 32u; ++mm) {
        uint32_t t = (r & 0x3FFFu) | ((32u * g + mm) << 14);
        double ua, ub, u1, u2;
        orc_uniform_pair(seed, pid, attempt, TAG(P_INIT, t, k), &ua, &ub);
        double x = sqrt(-2.0 * log(ua)) * cos(TWO_PI * ub);
        double v = 1.0 + c * x;
        if (!(v > 0.0)) continue;
        v = v * v * v;
        orc_uniform_pair(seed, pid, attempt, TAG(P_INIT, t, k | 64u), &u1, &u2);
        double x2 = x * x;
        if (u1 < 1.0 - 0.0331 * x2 * x2 || log(u1) < 0.5 * x2 + dd * (1.0 - v + log(v))) {
            double G = dd * v;
            if (boost) G *= exp(log(u2) / shape);
            return G;
        }
    }
    return NAN;
}
static double prior_draw(uint64_t seed, uint64_t pid, uint32_t attempt, uint32_t r, uint32_t k, int fam, double a, double b) {
    if (fam == ORC_PRIOR_NORMAL || fam == ORC_PRIOR_UNIFORM) {
        double ua, ub;
        orc_uniform_pair(seed, pid, attempt, TAG(P_INIT, r, k), &ua, &ub);
        return fam == ORC_PRIOR_NORMAL ? a + b * (sqrt(-2.0 * log(ua)) * cos(TWO_PI * ub)) : a + (b - a) * ua;
    }
    switch (fam) {
    case ORC_PRIOR_GAMMA: return b * gamma_unit_draw(seed, pid, attempt, r, k, 0u, a);
    case ORC_PRIOR_BETA: {
        double g1 = gamma_unit_draw(seed, pid, attempt, r, k, 0u, a), g2 = gamma_unit_draw(seed, pid, attempt, r, k, 1u, b);
        return g1 / (g1 + g2);
    }
    case ORC_PRIOR_INVGAMMA: return b / gamma_unit_draw(seed, pid, attempt, r, k, 0u, a);
    case ORC_PRIOR_ROOTINVGAMMA: return sqrt(a * b * b / (2.0 * gamma_unit_draw(seed, pid, attempt, r, k, 0u, 0.5 * a)));
    default: return NAN;
    }
}
int orc_initial_draw(const orc_model *m, double *particles, int64_t n, int64_t pid0, uint64_t seed) {
    int d = m->n_para, R = d + 5;
    if (d > 64) return fail("initial_draw: the RNG tags carry the parameter index in 6 bits");
    for (int64_t i = 0; i < n; ++i) {
        double th[ORC_MAXD], ll = 0, lp = 0;
        uint64_t pid = (uint64_t)(pid0 + i);
        for (uint32_t attempt = 0;; ++attempt) {
            if (attempt > 100000) return fail("initial_draw: no finite-likelihood draw after 100000 attempts");
            for (int k = 0; k < d; ++k) {
                if (m->fixed[k]) { th[k] = m->prior_a[k]; continue; }   /* fixed: value carried in prior_a */
                int fam = m->prior_family[k];
                uint32_t r_max = (fam == ORC_PRIOR_NORMAL || fam == ORC_PRIOR_UNIFORM) ? 100000u : 16383u;
                for (uint32_t r = 0;; ++r) {
                    double x = prior_draw(seed, pid, attempt, r, (uint32_t)k, fam, m->prior_a[k], m->prior_b[k]);
                    if (m->lo[k] < x && x < m->hi[k]) { th[k] = x; break; }  /* rand(::ParameterVector) redraw */
                    if (r >= r_max) return fail("initial_draw: prior draw never inside bounds");
                }
            }
            if (orc_in_bounds(m, th)) {
                ll = orc_loglik(&m->lik, th, d);
                lp = orc_logprior(m, th);
                if (ll == -INFINITY || isnan(ll)) ll = lp = -INFINITY;
            } else ll = lp = -INFINITY;
            if (!isinf(ll)) break;                                        /* initialization.jl:56-60 */
        }
        for (int k = 0; k < d; ++k) particles[(int64_t)k * n + i] = th[k];
        particles[(int64_t)d * n + i] = ll;
        particles[(int64_t)(d + 1) * n + i] = lp;
        particles[(int64_t)(d + 2) * n + i] = 0.0;                        /* update_old_loglh!(c, zeros) */
        particles[(int64_t)(d + 3) * n + i] = 0.0;
        particles[(int64_t)(R - 1) * n + i] = 1.0;                        /* set_weights!(c, ones) */
    }
    return 0;
}

/* ------------------------------------------------------------------ smc_main.jl:377-508 */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_smc_run(const orc_model *m, const orc_run_config *cfg, double *particles, double *sched_out,
                double *ess_out, double *c_out, double *accept_out, int32_t *resampled_out, double *w_hist,
                double *W_hist, orc_run_result *res) {
    const int64_t n = cfg->n_parts;
    const int d = m->n_para, R = d + 5, n_phi = cfg->n_phi;
    int free_inds[ORC_MAXD], n_free = 0;
    for (int k = 0; k < d; ++k) if (!m->fixed[k]) free_inds[n_free++] = k;
    if (n_free == 0) return fail("All model parameters are fixed!");
    if (cfg->n_blocks < 1 || (n_free + cfg->n_blocks - 1) / cfg->n_blocks * (cfg->n_blocks - 1) >= n_free)
        return fail("n_blocks incompatible with the number of free parameters");
    double *sched = (double *)malloc(sizeof(double) * n_phi);
    for (int k = 0; k < n_phi; ++k) sched[k] = pow((double)k / (double)(n_phi - 1), cfg->lambda); /* :348-352 */
    double *inc_w = (double *)malloc(sizeof(double) * n), *norm_w = (double *)malloc(sizeof(double) * n);
    double *tmp = (double *)malloc(sizeof(double) * n * R), *rw = (double *)malloc(sizeof(double) * n);
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * n);
    double *mean = (double *)malloc(sizeof(double) * d), *cov = (double *)malloc(sizeof(double) * d * d);
    double *mu_f = (double *)malloc(sizeof(double) * n_free), *Sig_f = (double *)malloc(sizeof(double) * n_free * n_free);
    int32_t bfree[ORC_MAXD], ball[ORC_MAXD], bptr[ORC_MAXD + 1];
    double *wcol = particles + (int64_t)(R - 1) * n, *acol = particles + (int64_t)(R - 2) * n;

    g_refactor_per_particle = cfg->variant == 1;
    /* variant 2: ONE team size for every parallel region of the run (libgomp rebuilds its pool when the size changes), capped so
       that a thread has a few thousand particles: at N = 1e5 waking 255 sleeping threads costs more than their share of the work */
    int mut_threads = cfg->n_threads;
    if (cfg->variant == 2) {
        const char *ot = getenv("ORC_OPT_THREADS");
        int64_t cap = ot ? atoi(ot) : n / 3072;      /* (32 threads at N = 1e5: measured best on the 256-thread host, 1.9 s vs 2.4 - 2.7 s at 48) */
        /* a Kalman-filter likelihood costs ~0.5 MFLOP per proposal: the mutation loop, not the wake-ups, is the run - a thread per 48
           particles (round 3 capped it at 4 threads for 12 500 particles and the "optimised" variant came out slower than the faithful one) */
        if (!ot && m->lik.family == ORC_LIK_LGSS_KALMAN && cap < n / 48) cap = n / 48;
        if (cap < 1) cap = 1;
        if ((int64_t)mut_threads > cap) mut_threads = (int)cap;
    }
    g_ess_threads = (cfg->variant == 2 && mut_threads > 1) ? mut_threads : 1;
    double t_solve = 0.0, t_corr = 0.0, t_sel = 0.0, t_mom = 0.0, t_mut = 0.0;
    const int profile = getenv("ORC_PROFILE") != NULL;
    int i = 1, j = 2, rc = 0, resampled_last = 0, resamples = 0;
    double phi_n = 0.0, phi_prop = 0.0, c = cfg->c, accept = cfg->target, logmdd = 0.0, secs = 0.0;
    const double threshold = cfg->threshold_ratio * (double)n;
    sched_out[0] = 0.0; ess_out[0] = cfg->initial_ess > 0.0 ? cfg->initial_ess : (double)n;   /* initialization.jl:199-200 */
    c_out[0] = c; accept_out[0] = accept; resampled_out[0] = 0;
    if (w_hist) for (int64_t k = 0; k < n; ++k) { w_hist[k] = 0.0; W_hist[k] = wcol[k]; }       /* :363-366 */

    while (phi_n < 1.0) {                                                                        /* :377 */
        double t0 = now_s();
        i += 1;
        if (i > cfg->max_stages) { rc = fail("max_stages exceeded"); break; }
        double phi_n1 = sched_out[i - 2];
        double tp = now_s();
        if (cfg->use_fixed_schedule) phi_n = sched[i - 1];                                       /* :387 */
        else if ((rc = orc_solve_adaptive_phi(particles, n, R, ess_out[i - 2], sched, n_phi, &j, &phi_prop, phi_n1,
                                              cfg->tempering_target, &resampled_last, &phi_n, NULL)) != 0) break;
        sched_out[i - 1] = phi_n;
        t_solve += now_s() - tp; tp = now_s();
        double ess, sum_un;
        orc_correct(particles, n, R, phi_n, phi_n1, cfg->prior_weight, cfg->log_prob_old_data, inc_w, norm_w, &ess, &sum_un);
        ess_out[i - 1] = ess;                                                                    /* :427 */
        logmdd += log(sum_un / (double)n);                                                       /* SURVEY a-9 */
        if (w_hist) { memcpy(w_hist + (int64_t)(i - 1) * n, inc_w, sizeof(double) * n);
                      memcpy(W_hist + (int64_t)(i - 1) * n, norm_w, sizeof(double) * n); }
        if (isnan(ess)) { rc = fail("No particles have non-zero weight."); break; }              /* :431 */
        resampled_out[i - 1] = 0;
        t_corr += now_s() - tp; tp = now_s();
        if (ess < threshold) {                                                                   /* :435-446 */
            for (int64_t k = 0; k < n; ++k) rw[k] = norm_w[k] / (double)n;
            orc_resample(rw, n, n, cfg->resampling_method, cfg->seed, (uint32_t)i, idx);
            for (int col = 0; col < R; ++col)
                for (int64_t k = 0; k < n; ++k) tmp[(int64_t)col * n + k] = particles[(int64_t)col * n + idx[k]];
            memcpy(particles, tmp, sizeof(double) * n * R);
            for (int64_t k = 0; k < n; ++k) wcol[k] = 1.0;
            resamples += 1; resampled_last = 1; resampled_out[i - 1] = 1;
            if (W_hist) for (int64_t k = 0; k < n; ++k) W_hist[(int64_t)(i - 1) * n + k] = 1.0;
        }
        t_sel += now_s() - tp; tp = now_s();
        c = orc_update_c(c, accept, cfg->target);                                                /* :453-455 */
        c_out[i - 1] = c;
        orc_weighted_mean(particles, n, R, mean);                                                /* :457-458 */
        orc_weighted_cov(particles, n, R, cov);
        for (int a = 0; a < n_free; ++a) {                                                       /* :462-465 */
            mu_f[a] = mean[free_inds[a]];
            for (int b = 0; b < n_free; ++b)
                Sig_f[a * n_free + b] = (cov[free_inds[a] * d + free_inds[b]] + cov[free_inds[b] * d + free_inds[a]]) / 2.0;
        }
        t_mom += now_s() - tp; tp = now_s();
        orc_generate_blocks(n_free, cfg->n_blocks, free_inds, cfg->seed, (uint32_t)i, bfree, ball, bptr); /* :468-469 */
        if ((rc = orc_mutate_cloud(m, particles, n, 0, mu_f, Sig_f, n_free, bfree, ball, bptr, cfg->n_blocks, phi_n,
                                   phi_n1, c, cfg->alpha, cfg->n_mh_steps, cfg->seed, (uint32_t)i, mut_threads)) != 0) break;
        double sa = 0.0;
        for (int64_t k = 0; k < n; ++k) sa += acol[k];
        accept = sa / (double)n;                                                                 /* :484 */
        accept_out[i - 1] = accept;
        t_mut += now_s() - tp;
        secs += now_s() - t0;                                                                    /* :489-490 */
    }
    if (profile) fprintf(stderr, "[orc] variant %d threads %d: solve %.3f correction %.3f selection %.3f moments %.3f mutation %.3f s\n", cfg->variant,
                         mut_threads, t_solve, t_corr, t_sel, t_mom, t_mut);
    res->n_stages = i; res->resamples = resamples; res->logmdd = logmdd; res->c = c; res->accept = accept;
    res->seconds = secs;
    g_refactor_per_particle = 0; g_ess_threads = 1;
    free(sched); free(inc_w); free(norm_w); free(tmp); free(rw); free(idx); free(mean); free(cov); free(mu_f); free(Sig_f);
    return rc;
}
