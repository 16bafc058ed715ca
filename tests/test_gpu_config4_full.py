"""BASELINE config 4 at its FULL size (examples/capm_model, N = 200 000, fixed schedule n_Φ = 300, 3 MH steps; src/mutation.jl:123-133,
examples/capm_model/estimate_capm.jl:52-70) against the oracle - VERDICT r3 weak 1.

What can and cannot agree.  The estimator is extremely noisy on this model (diffuse N(0, 1e3) / U(0, 1e3) priors): across Philox seeds its
log-MDD has a standard deviation of ≈ 5.5 at N = 200 000 (profiles/r04_capm_gap_n200000.json: 16 device seeds, 6 oracle seeds, means
-157.47 / -157.57).  Device and oracle start from the same cloud and agree to rounding stage after stage (ESS relative difference 2e-14 at
stage 6, growing ≈ 1.2x per stage through the adaptive proposal: every particle's proposal depends on the cloud's covariance) until the
drift - device exp / log against glibc's, blocked sums against serial ones - flips ONE Metropolis-Hastings decision (stage 66 of 300 on
seed 1: 600 000 decisions per stage); from there the two runs are different realisations of the same estimator and their log-MDDs differ
like two seeds do, only less (0.35 on seed 1).  At N = 10 000 (tests/test_gpu_parity.py::test_run_capm_config4_vs_oracle) the flip does not
happen within 300 stages and the runs agree to 1e-3.

So the full-size test asserts what a defect in the engine that serves this size (engine 1: n_para = 9 but N > 131 072) would break:
  * every bracketed stage - the run paused after stage k - 1 and after stage k, the oracle repeating exactly that stage on the downloaded
    cloud - has the oracle's ancestors, ZERO flipped decisions on the strict-FP build (≤ 3 on the product build), the oracle's weights to
    1e-11 and the oracle's values to rounding, early and late in the run, with and without a resample;
  * the two whole runs agree to rounding over a long prefix (≥ 40 stages with ESS within 1e-8 relative, the same resample stages, partial
    log-MDD sums within 1e-6) - a systematic error would part them at once;
  * the final log-MDDs are within the estimator's own spread."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRICT = os.path.join(ROOT, "smc.jl_amd", "csrc", "libsmcmi_strict.so")

_W = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle as orc
from smc_jl_amd import Engine
from tests import models
n, seed, stages = %(n)d, %(seed)d, %(stages)r
spec = models.capm_spec(); m = models.oracle_model(spec); d = 9
kw = dict(use_fixed_schedule=True, n_phi=300, lam=2.1, n_mh_steps=3)
e = Engine(n, d, seed=seed, max_stages=300, store_history=True)
e.set_model(spec); e.init_from_prior()
P0 = e.download_cloud()
g = e.run(**kw)
rec = e.stage_records(g["n_stages"])
w, Wn = e.history(g["n_stages"])
inc_g = np.log(np.sum(w[:, 1:] * Wn[:, :-1], axis=0) / n)
del w, Wn
r = orc.smc_run(m, P0, seed=seed, n_threads=64, history=True, **kw)
inc_c = np.log(np.sum(r["w"][:, 1:] * r["W"][:, :-1], axis=0) / n)
rel = np.abs(rec["ess"] - r["ess"]) / r["ess"]
part = np.nonzero(rel > 1e-8)[0]
k_part = int(part[0]) + 1 if part.size else 301                      # first stage whose ESS differs beyond rounding drift (the first flipped decision shows as ~1e-7)
out = dict(logmdd_gpu=g["logmdd"], logmdd_cpu=r["logmdd"], n_stages=[g["n_stages"], r["n_stages"]], k_part=k_part,
           prefix_inc_err=float(abs(np.sum(inc_g[:k_part - 2] - inc_c[:k_part - 2]))),
           prefix_resampled_equal=bool(np.array_equal(rec["resampled"][:k_part - 1], r["resampled"][:k_part - 1])),
           prefix_accept_err=float(np.max(np.abs(rec["accept_hist"][:k_part - 1] - r["accept_hist"][:k_part - 1]))),
           ess_rel_stage_6=float(rel[5]), ess_rel_stage_40=float(rel[39]), hist_formula_err=float(abs(np.sum(inc_g) - g["logmdd"])))
del r
# ---- bracketed stages of a second run of the same engine
e2 = Engine(n, d, seed=seed, max_stages=300, store_history=False)
e2.set_model(spec); e2.init_from_prior()
br, cont = [], False
for k in stages:
    e2.run(stop_after_stage=k - 1, continue_run=cont, **kw); cont = True
    A = e2.download_cloud()
    rr = e2.run(stop_after_stage=k, continue_run=True, **kw)
    B = e2.download_cloud()
    rc2 = e2.stage_records(rr["n_stages"])
    phi1, phi0, resampled, c = rc2["schedule"][k - 1], rc2["schedule"][k - 2], int(rc2["resampled"][k - 1]), rc2["c_hist"][k - 1]
    Pc, incw, nw, ess, su = orc.correct(A, phi1, phi0)                 # smc_main.jl:401-432
    res_cpu = int(ess < 0.5 * n)
    if res_cpu:                                                        # :435-446
        idx = orc.resample(Pc[:, d + 4] / n, "systematic", seed=seed, stage=k)
        Pc = np.asfortranarray(Pc[idx]); Pc[:, d + 4] = 1.0
    mean, cov = orc.weighted_mean(Pc), orc.weighted_cov(Pc)          # :457-465
    bf, ba, bp = orc.generate_blocks(d, 1, m.free_inds, seed, k)
    want = orc.mutate_cloud(m, Pc, mean, (cov + cov.T) / 2, bf, ba, bp, phi1, phi0, c, 1.0, 3, seed, k, n_threads=64)
    flips = int(np.count_nonzero(B[:, d + 3] != want[:, d + 3]))
    # values: relative to the element, plus the rounding of a proposal x + c L z whose terms (~ the column's scale) cancel
    scale = np.max(np.abs(want[:, :d + 3]), axis=0)
    bad = np.abs(B[:, :d + 3] - want[:, :d + 3]) > 1e-9 * (1 + np.abs(want[:, :d + 3])) + 1e-11 * scale
    br.append(dict(stage=k, resampled_gpu=resampled, resampled_cpu=res_cpu, ess_rel=float(abs(ess - rc2["ess"][k - 1]) / ess), flips=flips,
                   rows_differ=int(np.count_nonzero(np.any(bad, axis=1))),
                   w_rel=float(np.max(np.abs(B[:, d + 4] - want[:, d + 4]) / (1e-300 + np.abs(want[:, d + 4]))))))
out["bracket"] = br
print("RESULT " + json.dumps(out))
'''

STAGES = [2, 6, 13, 40, 66, 67, 105, 126, 150, 200, 258, 299]


def _run(lib, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    if lib:
        env["SMCMI_LIBRARY"] = lib
    code = _W % dict(root=ROOT, n=200000, seed=1, stages=STAGES)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=2400, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


@pytest.mark.parametrize("build", ["strict", "product"])
def test_config4_full_size_against_the_oracle(build):
    assert os.path.exists(STRICT), "libsmcmi_strict.so missing: python __graft_entry__.py builds it"
    o = _run(STRICT if build == "strict" else None)
    print("config 4, N = 200000, %s build: log-MDD device %.6f oracle %.6f; runs agree to rounding up to stage %d (partial log-MDD error %.2e); bracketed stages: %s"
          % (build, o["logmdd_gpu"], o["logmdd_cpu"], o["k_part"] - 1, o["prefix_inc_err"],
             ", ".join("%d:%d flips" % (b["stage"], b["flips"]) for b in o["bracket"])))
    assert o["n_stages"] == [300, 300]
    assert o["hist_formula_err"] < 1e-7                                  # log-MDD = Σ log((1/N) Σ w W) of the stored history
    # one stage at a time the engine IS the oracle, wherever in the run
    assert any(b["resampled_cpu"] for b in o["bracket"]) and not all(b["resampled_cpu"] for b in o["bracket"])
    for b in o["bracket"]:
        assert b["resampled_gpu"] == b["resampled_cpu"], b
        assert b["ess_rel"] < 1e-11 and b["w_rel"] < 1e-11, b
        assert b["flips"] <= (0 if build == "strict" else 3), b
        assert b["rows_differ"] <= 3 * b["flips"], b                     # same ancestors, same proposals, same values
    # the whole runs: rounding-level agreement over a long prefix (a systematic defect parts them in the first stages)
    assert o["k_part"] >= 40, o
    assert o["ess_rel_stage_6"] < 1e-11 and o["ess_rel_stage_40"] < 1e-8, o
    assert o["prefix_resampled_equal"] and o["prefix_inc_err"] < 1e-6 and o["prefix_accept_err"] < 1e-4, o
    # beyond the first flipped decision: two realisations of an estimator whose seed-to-seed sd is 5.5 on this workload
    assert abs(o["logmdd_gpu"] - o["logmdd_cpu"]) < 3.0, o
