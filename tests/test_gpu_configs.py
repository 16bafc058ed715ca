"""BASELINE configs at their full sizes on one MI355X, and the shard-count invariance SURVEY §8(e) asks of config 3:
the same seed must give the identical ϕ schedule, ancestors and cloud on 1, 2, 4 and 8 shards.

Engine 2 (csrc/stage2.hpp) totals every per-block quantity per *virtual shard* (8 fixed global particle ranges) in a canonical
order, so a handle that holds 8, 4, 2 or 1 virtual shards produces the same bits.  The in-process group driver runs exactly the
code of the RCCL driver with device copies instead of ncclAllGather / ncclSend / ncclRecv (one GPU is all a gpurun box has).
Every run here goes through the C ABI (smcmi_run / smcmi_run_group); the oracle is only the checker."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import models

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import json, os, sys, hashlib
import numpy as np
sys.path.insert(0, %(root)r)
from smc_jl_amd import Engine, run_group
from tests import models
n, d, seed, worlds, kw = %(n)d, %(d)d, %(seed)d, %(worlds)r, %(kw)r
spec_name, spec_args = %(spec)r
spec = getattr(models, spec_name)(*(spec_args if spec_args is not None else [d]))
out = {}
for world in worlds:
    nl = n // world
    engs = []
    for r in range(world):
        e = Engine(n, d, seed=seed, max_stages=1500, store_history=False, n_local=nl, gid0=r * nl)
        e.set_model(spec)
        e.init_from_prior()
        engs.append(e)
    res = run_group(engs, **kw) if world > 1 else engs[0].run(**kw)
    rec = engs[0].stage_records(res["n_stages"])
    cloud = np.concatenate([e.download_cloud() for e in engs], axis=0)
    anc = np.concatenate([e.last_ancestors() for e in engs]) if hasattr(engs[0], "last_ancestors") else np.zeros(0)
    out[str(world)] = dict(n_stages=res["n_stages"], resamples=res["resamples"], logmdd=res["logmdd"].hex() if hasattr(res["logmdd"], "hex") else float(res["logmdd"]).hex(),
                           schedule=hashlib.sha256(np.ascontiguousarray(rec["schedule"]).tobytes()).hexdigest(),
                           ess=hashlib.sha256(np.ascontiguousarray(rec["ess"]).tobytes()).hexdigest(),
                           accept=hashlib.sha256(np.ascontiguousarray(rec["accept_hist"]).tobytes()).hexdigest(),
                           cloud=hashlib.sha256(np.ascontiguousarray(cloud).tobytes()).hexdigest(),
                           mean0=float(cloud[:, 0].mean()), stalls=[res.get("solver_stalls", 0), res.get("select_stalls", 0), res.get("spec_stalls", 0)])
    for e in engs:
        e.close()
print("RESULT " + json.dumps(out))
'''


def _invariance(n, d, seed, worlds, kw, extra_env=None, spec=("gauss_spec", None)):
    env = dict(os.environ, SMCMI_ENGINE="2")          # world = 1 takes engine 2 as well (the default there is size-dependent)
    env.update(extra_env or {})
    code = _WORKER % dict(root=ROOT, n=n, d=d, seed=seed, worlds=list(worlds), kw=kw, spec=(spec[0], spec[1]))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("kw", [
    dict(use_fixed_schedule=False, tempering_target=0.95, n_blocks=2, alpha=0.9),
    dict(use_fixed_schedule=True, n_phi=60, n_mh_steps=2),
    dict(use_fixed_schedule=False, tempering_target=0.97, resampling_method="multinomial"),
])
def test_results_do_not_depend_on_the_shard_count(kw):
    """1, 2, 4 and 8 in-process shards of one population: bit-identical schedule, ESS path, acceptance rates, log-MDD and cloud
    (hence identical ancestors on every resample stage: a differing ancestor would show in the cloud)."""
    out = _invariance(40000, 6, 13, (1, 2, 4, 8), kw)
    ref = out["1"]
    assert ref["resamples"] >= 2 and ref["n_stages"] > 10
    for w in ("2", "4", "8"):
        for key in ("n_stages", "resamples", "logmdd", "schedule", "ess", "accept", "cloud"):
            assert out[w][key] == ref[key], (w, key, out[w], ref)


@pytest.mark.parametrize("n,worlds,kw", [
    (1_000_000, (1, 2, 4), dict(use_fixed_schedule=False, tempering_target=0.97)),
    (400_000, (1, 2), dict(use_fixed_schedule=True, n_phi=40, n_mh_steps=2, n_blocks=2, alpha=0.9)),
    (400_000, (1, 2), dict(use_fixed_schedule=False, tempering_target=0.9, resampling_method="multinomial")),
])
def test_large_shards_do_not_depend_on_the_shard_count(n, worlds, kw):
    """Shards of more than 131 072 particles (what a rank of a 2- or 4-GPU run of config 3 holds: 500 000 / 250 000) run engine 2's
    large-shard stage (csrc/stage2b.hpp: k2b_mutate at two 512-particle blocks per CU, rows totalled per virtual shard in the canonical
    order).  One handle takes it with the self-addressed mailbox and the helper blocks (decision + proposal inside K1's launch, the next
    stage's begin inside the mutation launch); the in-process groups hand their totals over with device copies and run k2_begin /
    k2_prepare as launches - the bits must be the same on 1, 2 and 4 shards either way (src/smc_main.jl:472-476: the reference's
    particle loop does not depend on the number of workers either)."""
    d = 10
    out = _invariance(n, d, 3, worlds, kw)
    ref = out["1"]
    assert ref["n_stages"] > 10 and ref["resamples"] >= 1
    for w in worlds[1:]:
        for key in ("n_stages", "resamples", "logmdd", "schedule", "ess", "accept", "cloud"):
            assert out[str(w)][key] == ref[key], (w, key, out[str(w)], ref)
    # two in-process handles with the peer mailbox: helper blocks on both, each polling for the other's totals
    out_mb = _invariance(n, d, 3, (2,), kw, extra_env={"SMCMI_MAILBOX": "1"})
    for key in ("n_stages", "resamples", "logmdd", "schedule", "ess", "accept", "cloud"):
        assert out_mb["2"][key] == ref[key], ("mailbox", key, out_mb["2"], ref)
    # ... and one handle with k2_begin / k2_prepare as launches (SMCMI_E2_HELPERS=0)
    out_nh = _invariance(n, d, 3, (1,), kw, extra_env={"SMCMI_E2_HELPERS": "0"})
    for key in ("n_stages", "resamples", "logmdd", "schedule", "ess", "accept", "cloud"):
        assert out_nh["1"][key] == ref[key], ("no helpers", key, out_nh["1"], ref)


@pytest.mark.parametrize("case", [
    dict(spec=("gauss_spec", [12]), n=24000, d=12, kw=dict(use_fixed_schedule=False, tempering_target=0.95, n_blocks=3, alpha=0.9), env={}),
    dict(spec=("gauss_spec", [16]), n=16384, d=16, kw=dict(use_fixed_schedule=True, n_phi=40, n_mh_steps=2, resampling_method="multinomial"), env={}),
    dict(spec=("kalman_spec", [40]), n=16384, d=13, kw=dict(use_fixed_schedule=False, tempering_target=0.95, n_phi=100, alpha=0.9), env={}),              # four lanes per particle
    dict(spec=("kalman_spec", [40]), n=16384, d=13, kw=dict(use_fixed_schedule=False, tempering_target=0.95, n_phi=100, alpha=0.9), env={"SMCMI_KALMAN_LANES": "1"}),
], ids=["gauss12", "gauss16_fixed_multinomial", "kalman_quad", "kalman_one_thread"])
def test_models_with_11_to_16_parameters_do_not_depend_on_the_shard_count(case):
    """n_para > 10 (VERDICT r3 missing 2: the reference's loop has no dimension cliff, src/smc_main.jl:207-236) runs the same two-launch
    stage as the small models - K1 with the row's sums formed sixteen at a time, K2's prologue in front of the generic mutation body
    (stage2.hpp k2w_mutate) - so 1, 2 and 4 shards give the same bits: schedule, ESS path, acceptance rates, log-MDD, cloud.  (The two
    Kalman filters - four lanes per particle on small clouds, one thread per particle beyond - sum in different orders: the invariance
    holds per filter, which the handle picks by ITS cloud size unless SMCMI_KALMAN_LANES says otherwise.)"""
    out = _invariance(case["n"], case["d"], 7, (1, 2, 4), case["kw"], extra_env=case["env"], spec=case["spec"])
    ref = out["1"]
    assert ref["n_stages"] > 10 and ref["resamples"] >= 1
    for w in ("2", "4"):
        for key in ("n_stages", "resamples", "logmdd", "schedule", "ess", "accept", "cloud"):
            assert out[w][key] == ref[key], (w, key, out[w], ref)


def test_the_wide_stage_against_engine_1s():
    """The same 12- and 13-parameter runs through engine 1's eight-launch stage (SMCMI_ENGINE=1; sums in another order): same stage
    and resample counts, log-MDD to 1e-7."""
    for spec, n, d, kw in [(("gauss_spec", [12]), 24000, 12, dict(use_fixed_schedule=False, tempering_target=0.95, n_blocks=3, alpha=0.9)),
                           (("kalman_spec", [40]), 16384, 13, dict(use_fixed_schedule=False, tempering_target=0.95, n_phi=100, alpha=0.9))]:
        a = _invariance(n, d, 7, (1,), kw, spec=spec)["1"]
        b = _invariance(n, d, 7, (1,), kw, spec=spec, extra_env={"SMCMI_ENGINE": "1"})["1"]
        assert a["n_stages"] == b["n_stages"] and a["resamples"] == b["resamples"], (a, b)
        assert float.fromhex(a["logmdd"]) == pytest.approx(float.fromhex(b["logmdd"]), abs=1e-7)


def test_direct_and_reduced_geometry_agree_bitwise():
    """One handle: every block totalling the per-block rows itself (small clouds) and the k2_reduce / one-block set-up path used
    for large clouds and shards are two implementations of the same canonical order."""
    kw = dict(use_fixed_schedule=False, tempering_target=0.95, n_blocks=2, alpha=0.9)
    a = _invariance(40000, 6, 13, (1,), kw)["1"]
    b = _invariance(40000, 6, 13, (1,), kw, extra_env={"SMCMI_E2_REDUCED": "1"})["1"]
    for key in ("n_stages", "resamples", "logmdd", "schedule", "ess", "accept", "cloud"):
        assert a[key] == b[key], key


def test_config3_workload_on_eight_shards_equals_one_handle():
    """BASELINE config 3's workload (10-dim Gaussian, adaptive ϕ, 0.97) sharded over 8 handles of one process vs one handle: the
    same bits, the analytic log-MDD within Monte-Carlo error."""
    n = 400000                        # 50 000 per shard (config 3 itself is 125 000 per GPU: bench.py --gpus 8)
    kw = dict(use_fixed_schedule=False, tempering_target=0.97, n_phi=300)
    out = _invariance(n, 10, 1, (1, 8), kw)
    for key in ("n_stages", "resamples", "logmdd", "schedule", "ess", "accept", "cloud"):
        assert out["8"][key] == out["1"][key], key
    assert abs(float.fromhex(out["1"]["logmdd"]) - models.gauss_logmdd(10)) < 0.1
    assert abs(out["1"]["mean0"] - (-1.0) * 25 / 25.0625) < 0.01


def test_config3_in_its_own_shape_eight_shards_of_125000():
    """BASELINE config 3 itself: N = 1 000 000 as 8 x 125 000 (the per-GPU shard of `bench.py --gpus 8`), 8 handles of one process
    against one handle of 10⁶ particles on the same engine: bit-identical, analytic log-MDD within Monte-Carlo error."""
    kw = dict(use_fixed_schedule=False, tempering_target=0.97, n_phi=300)
    out = _invariance(1_000_000, 10, 1, (1, 8), kw)
    for key in ("n_stages", "resamples", "logmdd", "schedule", "ess", "accept", "cloud"):
        assert out["8"][key] == out["1"][key], key
    assert out["1"]["resamples"] >= 8
    assert abs(float.fromhex(out["1"]["logmdd"]) - models.gauss_logmdd(10)) < 0.05
    assert abs(out["1"]["mean0"] - (-1.0) * 25 / 25.0625) < 0.005


def test_config2_workload_at_one_million_particles():
    """Config 2's workload at 10x its size on one GPU against the CPU oracle on the same Philox streams: identical stage and
    resample counts, ϕ and ESS paths to 1e-9, log-MDD to 1e-9 (the tolerance north_star states is 1e-3)."""
    from oracle import oracle as orc
    from smc_jl_amd import Engine

    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    n = 1_000_000
    spec = models.gauss_spec(10)
    eng = Engine(n, 10, seed=1, max_stages=1500, store_history=False)
    eng.set_model(spec)
    eng.init_from_prior()
    P0 = eng.download_cloud()
    kw = dict(use_fixed_schedule=False, tempering_target=0.97, n_phi=300, lam=2.1)
    r = eng.run(**kw)
    rec = eng.stage_records(r["n_stages"])
    P = eng.download_cloud()
    eng.close()
    # size-independent properties
    assert rec["schedule"][0] == 0.0 and rec["schedule"][-1] == 1.0 and np.all(np.diff(rec["schedule"]) > 0)
    assert np.all(rec["ess"] > 0) and np.all(rec["ess"] <= n * (1 + 1e-12))
    assert abs(r["logmdd"] - models.gauss_logmdd(10)) < 0.05
    w = P[:, -1]
    assert abs(w.sum() - n) < 1e-6 * n
    assert abs((P[:, 0] * w).sum() / w.sum() - (-1.0) * 25 / 25.0625) < 0.005
    ro = orc.smc_run(models.oracle_model(spec), P0, seed=1, n_threads=os.cpu_count(), history=False, max_stages=1500, **kw)
    assert r["n_stages"] == ro["n_stages"] and r["resamples"] == ro["resamples"]
    np.testing.assert_allclose(rec["schedule"], ro["schedule"], rtol=1e-9)
    np.testing.assert_allclose(rec["ess"], ro["ess"], rtol=1e-9)
    assert abs(r["logmdd"] - ro["logmdd"]) < 1e-9


def test_config5_at_fifty_thousand_particles_and_on_four_shards():
    """Config 5 (13-parameter state-space model, Kalman-filter likelihood, old 40 -> new 80 periods) at its full N = 50 000:
    properties of the run, and 4 in-process shards against the single handle."""
    from smc_jl_amd import Engine, run_group

    sp = models.kalman_spec(T=80, old_T=40)
    n = 50_000
    kw = dict(n_phi=100, use_fixed_schedule=False, tempering_target=0.95, n_blocks=1, alpha=0.9)
    e1 = Engine(n, 13, seed=1, max_stages=600, store_history=False)
    e1.set_model(sp)
    e1.init_from_prior()
    P0 = e1.download_cloud()
    r1 = e1.run(**kw)
    rec = e1.stage_records(r1["n_stages"])
    P1 = e1.download_cloud()
    e1.close()
    assert rec["schedule"][-1] == 1.0 and np.all(np.diff(rec["schedule"]) > 0)
    assert np.all(np.isfinite(P1)) and r1["resamples"] >= 1
    w = P1[:, -1]
    assert abs(w.sum() - n) < 1e-6 * n
    mu = (P1[:, :13] * w[:, None]).sum(0) / w.sum()
    assert abs(mu[12] - 1.0) < 0.5                       # measurement mean of the data-generating process
    assert np.all(P1[:, :8] >= -0.95) and np.all(P1[:, :8] <= 0.95)
    shards = []
    for k in range(4):
        e = Engine(n, 13, seed=1, n_local=n // 4, gid0=k * (n // 4), max_stages=600, store_history=False)
        e.set_model(sp)
        e.upload_cloud(P0[k * (n // 4):(k + 1) * (n // 4)])
        shards.append(e)
    r4 = run_group(shards, **kw)
    P4 = np.vstack([e.download_cloud() for e in shards])
    for e in shards:
        e.close()
    assert r1["n_stages"] == r4["n_stages"] and r1["resamples"] == r4["resamples"]
    # (the single handle of 50 000 runs the one-thread filter, the shards of 12 500 the four-lane one: the likelihoods agree to 1e-12, not
    # bitwise, so a handful of MH decisions may flip; with the same filter on both sides the bits are equal:
    # test_models_with_11_to_16_parameters_do_not_depend_on_the_shard_count)
    assert r4["logmdd"] == pytest.approx(r1["logmdd"], abs=1e-7)
    same = np.all(np.abs(P1 - P4) <= 1e-8 * (1 + np.abs(P1)), axis=1)
    assert same.mean() > 0.99
