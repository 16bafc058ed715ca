"""Host-side scalar logic shared by the multi-GPU orchestrator and the host-callback driver.

Everything here is O(d²) or O(K) work that the single-GPU path does in 1-block kernels
(csrc/kernels.hpp: k_stage_begin, solver_decide, k_post_correct, k_prepare_mutation); the sharded path
runs it on every rank from identical all-reduced inputs.  Same algorithms, same RNG contract.
"""
import math

import numpy as np

MASK32 = 0xFFFFFFFF
P_MUT, P_RES, P_BLK, P_INIT = 0, 1, 2, 3
KC = 16
RING = [2.0 ** -3, 2.0 ** -7, 2.0 ** -11, 2.0 ** -15, 2.0 ** -19, 2.0 ** -23]


def rng_tag(purpose, t, q):
    return ((purpose << 28) | ((t & 0xFFFFF) << 8) | (q & 0xFF)) & MASK32


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        n0 = ((p1 >> 32) ^ c1 ^ k0) & MASK32
        n2 = ((p0 >> 32) ^ c3 ^ k1) & MASK32
        c0, c1, c2, c3 = n0, p1 & MASK32, n2, p0 & MASK32
        k0 = (k0 + 0x9E3779B9) & MASK32
        k1 = (k1 + 0xBB67AE85) & MASK32
    return c0, c1, c2, c3


def uniform_pair(seed, pid, stage, tag):
    o = philox4x32_10(pid & MASK32, (pid >> 32) & MASK32, stage & MASK32, tag, seed & MASK32, (seed >> 32) & MASK32)
    ua = (float(((o[0] << 32) | o[1]) >> 11) + 0.5) * 2.0 ** -53
    ub = (float(((o[2] << 32) | o[3]) >> 11) + 0.5) * 2.0 ** -53
    return ua, ub


def schedule(n_phi, lam):
    """Proposed fixed schedule ((k-1)/(n_Φ-1))^λ, src/smc_main.jl:348-352."""
    return np.array([math.pow(k / (n_phi - 1.0), lam) for k in range(n_phi)])


def update_c(c, accept, target):
    """src/smc_main.jl:453-455."""
    e = math.exp(16.0 * (accept - target))
    return c * (0.95 + 0.10 * e / (1.0 + e))


def generate_blocks(n_free, n_blocks, free_inds, seed, stage):
    """generate_free_blocks / generate_all_blocks (src/helpers.jl:215-260): Fisher-Yates on Philox tag(P_BLK, i, 0)."""
    bf = list(range(n_free))
    for i in range(n_free - 1, 0, -1):
        ua, _ = uniform_pair(seed, 0, stage, rng_tag(P_BLK, i, 0))
        j = min(int(ua * (i + 1)), i)
        bf[i], bf[j] = bf[j], bf[i]
    sub = -(-n_free // n_blocks)
    bp = [b * sub for b in range(n_blocks)] + [n_free]
    bf = np.array(bf, dtype=np.int32)
    return bf, np.asarray(free_inds, dtype=np.int32)[bf], np.array(bp, dtype=np.int32)


def moments_from_totals(totals, shift, d):
    """Augmented pair sums Σ w x̃ x̃ᵀ, x̃ = (1, θ - shift) -> weighted_mean, weighted_cov (src/particle.jl:481-483,526-529)."""
    da = d + 1
    sw = totals[0]
    m1 = totals[1:da] / sw
    mean = np.asarray(shift) + m1
    cov = np.empty((d, d))
    for a in range(d):
        for b in range(a, d):
            ra, rb = a + 1, b + 1
            p = ra * da - ra * (ra - 1) // 2 + (rb - ra)
            cov[a, b] = cov[b, a] = totals[p] / sw - m1[a] * m1[b]
    return mean, cov


class PhiSolver:
    """solve_adaptive_ϕ (src/helpers.jl:9-56) as the same scan + secant/ring bracketing search the device runs
    (csrc/kernels.hpp solver_decide / section_candidates).  `ess_sums(cands) -> (Σv[k], Σv²[k])` evaluates a batch."""

    def __init__(self, sched, rtol=1e-12):
        self.sched, self.n_phi, self.rtol = np.asarray(sched, dtype=np.float64), len(sched), rtol

    def _section(self, lo, hi, glo, ghi):
        h = hi - lo
        out = []
        if h > self.rtol * hi:
            t = glo / (glo - ghi) if (glo > ghi and glo < 1e300 and ghi > -1e300) else 0.5
            xs = lo + h * t
            raw = [xs - h * r for r in RING] + [xs] + [xs + h * r for r in reversed(RING)]
            uni = [lo + h * (0.25 * u) for u in (1, 2, 3)]
            prev, u = lo, 0
            for x in raw:
                while u < 3 and uni[u] < x:
                    if prev < uni[u] < hi and len(out) < KC:
                        out.append(uni[u]); prev = uni[u]
                    u += 1
                if prev < x < hi and len(out) < KC:
                    out.append(x); prev = x
            while u < 3:
                if prev < uni[u] < hi and len(out) < KC:
                    out.append(uni[u]); prev = uni[u]
                u += 1
        return out

    def solve(self, ess_sums, j, phi_prop, phi_prev, ess_bar, ess_now, max_passes=64):
        lo, glo, hi, ghi = phi_prev, ess_now - ess_bar, phi_prop, 0.0
        cands = [phi_prop] + [self.sched[jj - 1] for jj in range(j, min(self.n_phi, j + KC - 2) + 1)]
        mode, passes = "scan", 0
        while True:
            passes += 1
            if passes > max_passes:
                raise RuntimeError("adaptive tempering solver did not converge")
            s1, s2 = ess_sums(np.asarray(cands, dtype=np.float64))
            g = np.asarray(s1) ** 2 / np.asarray(s2) - ess_bar
            m = next((k for k in range(len(cands)) if not g[k] >= 0.0), None)
            if mode == "scan":
                for k in range(len(cands) if m is None else m):
                    if cands[k] > lo:
                        lo, glo = cands[k], g[k]
                if m is not None:
                    if math.isnan(g[m]):
                        raise FloatingPointError("No particles have non-zero weight (ESS is NaN)")
                    phi_prop, j, hi, ghi, mode = cands[m], j + m, cands[m], g[m], "section"
                else:
                    j_new = j + len(cands) - 1
                    phi_prop = cands[-1]
                    if j_new > self.n_phi:
                        return phi_prop, j_new, phi_prop, passes        # ϕ_n = 1 (helpers.jl:51-53)
                    cands = [self.sched[jj - 1] for jj in range(j_new, min(self.n_phi, j_new + KC - 1) + 1)]
                    j, phi_prop = j_new + 1, cands[0]
                    continue
            else:
                if m is not None:
                    hi, ghi = cands[m], g[m]
                    if m > 0:
                        lo, glo = cands[m - 1], g[m - 1]
                else:
                    lo, glo = cands[-1], g[-1]
            cands = self._section(lo, hi, glo, ghi)
            if not cands:
                return (lo if abs(glo) <= abs(ghi) else hi), j, phi_prop, passes
