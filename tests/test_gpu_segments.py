"""Engine 3 (csrc/stage3.hpp): runs of stages that neither resample nor need a certificate pass execute as ONE persistent launch -
particles in registers, the two hand-overs of a stage as tickets + tagged records.  It calls engine 2's own row / decision / proposal /
MH functions on the same 512-particle blocks, so a run with segments must leave the bits a run of launches leaves (SMCMI_ENGINE3=0):
ϕ schedule, ESS path, acceptance rates, c, log-MDD, cloud, history.  Oracle parity of the default path is what every other GPU test
checks (they all run through it when the cloud is small enough)."""
import hashlib
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
import json, sys, hashlib
import numpy as np
sys.path.insert(0, %(root)r)
from smc_jl_amd import Engine
from tests import models
cfg = json.loads(%(cfg)r)
spec = getattr(models, cfg.get("spec", "gauss_spec"))(*cfg.get("spec_args", []))
e = Engine(cfg["n"], cfg["d"], seed=cfg["seed"], max_stages=cfg.get("max_stages", 1500), store_history=cfg.get("history", True))
e.set_model(spec)
out = []
P_old = None
if cfg.get("old_run"):                      # a tempered update: the cloud of an estimation on the old data is the starting point (smc_main.jl:244-260)
    e0 = Engine(cfg["n"], cfg["d"], seed=cfg["seed"] + 1, max_stages=200, store_history=False)
    e0.set_model(getattr(models, cfg["spec"])(*cfg["old_run"]["spec_args"])); e0.init_from_prior()
    e0.run(**cfg["old_run"]["kw"])
    P_old = e0.download_cloud(); e0.close()
for rep in range(cfg.get("reps", 1)):
    if P_old is None:
        e.init_from_prior()
    else:
        e.upload_cloud(P_old); e.initialize_likelihoods()
    kw = dict(cfg["kw"]); stop = kw.pop("pause_at", 0)
    if stop:
        r = e.run(stop_after_stage=stop, **kw); assert r["paused"]
        r = e.run(continue_run=True, **kw)
    else:
        r = e.run(**kw)
    rec = e.stage_records(r["n_stages"])
    h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    o = dict(n_stages=r["n_stages"], resamples=r["resamples"], logmdd=float(r["logmdd"]).hex(), c=float(r["c"]).hex(), accept=float(r["accept"]).hex(),
             schedule=h(rec["schedule"]), ess=h(rec["ess"]), c_hist=h(rec["c_hist"]), accept_hist=h(rec["accept_hist"]), resampled=h(rec["resampled"]),
             cloud=h(e.download_cloud()), n_segments=r["n_segments"], segment_stages=r["segment_stages"],
             stalls=[r["solver_stalls"], r["select_stalls"], r["spec_stalls"]], logmdd_f=r["logmdd"],
             shift_fallback_stage=r["shift_fallback_stage"], segment_blocks=r["segment_blocks"], segment_state=r["segment_state"],
             segment_timeouts=r["segment_timeouts"])
    if cfg.get("history", True):
        w, W = e.history(r["n_stages"])
        o["w"], o["W"] = h(w), h(W)
    out.append(o)
print("RESULT " + json.dumps(out))
'''


def _run(cfg, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    code = _WORKER % dict(root=ROOT, cfg=json.dumps(cfg))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


_KEYS = ("n_stages", "resamples", "logmdd", "c", "accept", "schedule", "ess", "c_hist", "accept_hist", "resampled", "cloud", "w", "W")


@pytest.mark.parametrize("cfg", [
    dict(n=100_000, d=10, seed=1, kw=dict(use_fixed_schedule=False, tempering_target=0.97), reps=2),                       # BASELINE config 2
    dict(n=20_000, d=10, seed=4, kw=dict(use_fixed_schedule=False, tempering_target=0.95, alpha=0.9, n_blocks=2, n_mh_steps=2)),
    dict(n=30_000, d=4, seed=2, spec_args=[4], kw=dict(use_fixed_schedule=True, n_phi=80, resampling_method="multinomial")),
    dict(n=5_000, d=2, seed=9, spec="regression_spec", kw=dict(use_fixed_schedule=False, tempering_target=0.9, pause_at=6)),
    dict(n=122_880, d=10, seed=3, kw=dict(use_fixed_schedule=False, tempering_target=0.97), history=False),                # 240 workers + 8 gatherers: nearly every CU
    dict(n=24_000, d=9, seed=6, spec="capm_spec", kw=dict(use_fixed_schedule=True, n_phi=120, n_mh_steps=3)),              # config 4's model: Uniform priors with bounds, 3 MH steps
    dict(n=16_000, d=9, seed=8, spec="linmodel_spec", spec_args=[100, 60], old_run=dict(spec_args=[60], kw=dict(n_phi=100, lam=2.0, alpha=0.9)),
         kw=dict(use_fixed_schedule=False, tempering_target=0.95, n_blocks=3, alpha=0.9)),                                  # a tempered update: an old-data likelihood beside the new one
    dict(n=16_000, d=9, seed=8, spec="linmodel_spec", spec_args=[100, 60], old_run=dict(spec_args=[60], kw=dict(n_phi=100, lam=2.0, alpha=0.9)),
         kw=dict(use_fixed_schedule=True, n_phi=60, lam=2.0, alpha=0.9, prior_weight=0.5, log_prob_old_data=-300.0)),        # mixed prior weight in the correction
    dict(n=50_002, d=10, seed=5, kw=dict(use_fixed_schedule=False, tempering_target=0.95), history=False),                 # two virtual shards of 49 rows: a gatherer's sweep in two batches
    dict(n=100_001, d=10, seed=7, kw=dict(use_fixed_schedule=False, tempering_target=0.95), history=False),                # no divisor among 8 / 4 / 2: virtual shards of ceil(n / 8), the last one shorter
], ids=["config2", "mix2blocks2steps", "fixed_multinomial", "regression_pause", "n122880", "capm3steps", "two_likelihoods", "prior_weight", "two_long_shards",
        "uneven_shards"])
def test_segments_leave_the_bits_of_the_launches(cfg):
    seg = _run(cfg)
    ref = _run(cfg, {"SMCMI_ENGINE3": "0"})
    for a, b in zip(seg, ref):
        assert b["n_segments"] == 0 and a["n_segments"] >= 1, (a["n_segments"], b["n_segments"])
        if cfg.get("spec") != "regression_spec":                                 # (its predictions rarely verify: heavy-tailed energies, target 0.9)
            assert a["segment_stages"] >= (a["n_stages"] - 1) // 2               # most stages ran inside segments
        for k in _KEYS:
            if k in b:
                assert a[k] == b[k], (k, a[k], b[k], a["stalls"], b["stalls"])
    if cfg.get("spec", "gauss_spec") == "gauss_spec" and cfg["d"] == 10 and cfg["kw"].get("alpha", 1.0) == 1.0:
        assert abs(seg[0]["logmdd_f"] - models.gauss_logmdd(10)) < 0.2


def test_a_mispredicted_resample_leaves_the_segment_and_is_redone():
    """SMCMI_NO_SELECT_PREDICT=2: the host expects no stage to resample, so every resample stage is met INSIDE a segment.  Where the segment
    cannot resample in place (mixture proposals, several handles; here: SMCMI_SEG_SELECT=0) it leaves with nothing of the stage committed
    (status 6); the host resumes the stage - correction and selection as launches, a segment entering at its mutation."""
    cfg = dict(n=40_000, d=6, seed=13, spec_args=[6], kw=dict(use_fixed_schedule=False, tempering_target=0.95))
    a = _run(cfg, {"SMCMI_NO_SELECT_PREDICT": "2", "SMCMI_SEG_SELECT": "0"})[0]
    b = _run(cfg, {"SMCMI_NO_SELECT_PREDICT": "2", "SMCMI_ENGINE3": "0"})[0]
    assert a["resamples"] >= 2 and a["stalls"][1] >= a["resamples"] - 1
    for k in _KEYS:
        assert a[k] == b[k], (k, a[k], b[k])


@pytest.mark.parametrize("kw,d", [(dict(use_fixed_schedule=True, n_phi=120), 7), (dict(use_fixed_schedule=False, tempering_target=0.95), 7),
                                  (dict(use_fixed_schedule=True, n_phi=80, resampling_method="multinomial", n_blocks=2, n_mh_steps=2), 7),
                                  (dict(use_fixed_schedule=False, tempering_target=0.95, alpha=0.9, n_blocks=2), 7),
                                  (dict(use_fixed_schedule=False, tempering_target=0.95, alpha=0.9), 10),
                                  (dict(use_fixed_schedule=True, n_phi=100, alpha=0.9, n_blocks=3), 9)],
                         ids=["fixed", "adaptive", "fixed_multinomial_2blocks", "adaptive_mixture_2blocks", "adaptive_mixture_d10", "fixed_mixture_d9_3blocks"])
def test_selection_inside_the_segment_leaves_the_bits_of_the_selection_launches(kw, d):
    """One handle: a stage that must resample does so inside the segment (stage3.hpp k3_select_inside: the workers scan their
    weights, find their ancestors and total the resampled cloud's moments with k2_scan's / k2_gather's own functions) - no stall, a run of
    one or two launches, and the bits of a run whose segments leave for the selection launches (SMCMI_SEG_SELECT=0) and of a run of launches.
    Mixture proposals beyond n_para 7 (the reference's own test configuration: 9 parameters, α = .9, test/smc.jl:26-29): the kernel's LDS has
    no room for the particle in transit - it is parked in the block's slice of a device buffer (Sel3Args::transit) - same bits, same launch count."""
    cfg = dict(n=60_000, d=d, seed=17, spec_args=[d], kw=kw)
    a = _run(cfg)[0]
    b = _run(cfg, {"SMCMI_SEG_SELECT": "0"})[0]
    c = _run(cfg, {"SMCMI_ENGINE3": "0"})[0]
    assert a["resamples"] >= 3 and a["stalls"][1] == 0 and a["n_segments"] <= 4 and b["n_segments"] > a["resamples"]
    for k in _KEYS:
        assert a[k] == b[k] == c[k], (k, a[k], b[k], c[k])


def test_fixed_schedules_take_one_hand_over_per_stage_and_the_lagged_shift_is_neutral():
    """Fixed schedules (the reference's default, src/smc_main.jl:139): the energy shift of a stage's incremental weights lags the cloud's largest
    energy by one mutation (RunParams::shift_lag), so that inside a segment a stage's correction row rides the mutation row in front of it - ONE
    hand-over per stage (stage3.hpp k3_rides).  The rule belongs to the run: launches (SMCMI_ENGINE3=0) leave the same bits - the parametrised
    tests above and below compare them on five fixed-schedule configurations.  Here: the shift itself is neutral - against exact shifts
    (SMCMI_SHIFT_LAG=0) the run has the same stages and resample decisions and its log-MDD, schedule and ESS path agree to rounding - and no
    fallback was needed."""
    cfg = dict(n=100_000, d=10, seed=1, kw=dict(use_fixed_schedule=True, n_phi=300))
    a = _run(cfg)[0]
    b = _run(cfg, {"SMCMI_SHIFT_LAG": "0"})[0]
    c = _run(cfg, {"SMCMI_ENGINE3": "0"})[0]
    assert a["n_segments"] == 1 and a["segment_stages"] == 299 and a["shift_fallback_stage"] == 0 and a["resamples"] >= 5
    assert a["segment_state"] == 1 and a["segment_blocks"] == 8 * ((12_500 + 511) // 512) + 8 and a["segment_timeouts"] == 0
    assert (a["n_stages"], a["resamples"], a["resampled"]) == (b["n_stages"], b["resamples"], b["resampled"])
    assert abs(a["logmdd_f"] - b["logmdd_f"]) <= 1e-9 * abs(b["logmdd_f"]), (a["logmdd_f"], b["logmdd_f"])
    assert abs(a["logmdd_f"] - models.gauss_logmdd(10)) < 0.2
    for k in _KEYS:
        assert a[k] == c[k], (k, a[k], c[k])


def test_sums_that_overflow_under_the_lagged_shift_switch_the_run_to_exact_shifts():
    """A lagged shift does not bound the weights by 1: a cloud whose largest log-likelihood grows by more than ~350 / (ϕ_n - ϕ_{n-1}) in ONE
    mutation overflows Σ W̃² at the next stage (decide2 counts sums that are not finite as check_nan_ess's case).  Nothing of that stage is
    committed; the host switches the run to exact shifts from that stage on (smcmi_result::shift_fallback_stage) - segments and launches
    alike, same bits - and the result is the run exact shifts give from the start: same stages, same resample decisions, log-MDD to rounding.
    SMCMI_SHIFT_LAG=12 (development) lowers stage 12's lagged shift by 1e6: the sums of that stage do overflow."""
    cfg = dict(n=20_000, d=10, seed=3, kw=dict(use_fixed_schedule=True, n_phi=60))
    a = _run(cfg, {"SMCMI_SHIFT_LAG": "12"})[0]
    b = _run(cfg, {"SMCMI_SHIFT_LAG": "0"})[0]
    c = _run(cfg, {"SMCMI_SHIFT_LAG": "12", "SMCMI_ENGINE3": "0"})[0]
    assert a["shift_fallback_stage"] == 12 and b["shift_fallback_stage"] == 0 and c["shift_fallback_stage"] == 12
    assert a["n_segments"] >= 2 and c["n_segments"] == 0
    assert (a["n_stages"], a["resamples"], a["resampled"]) == (b["n_stages"], b["resamples"], b["resampled"]) and a["n_stages"] == 60
    assert abs(a["logmdd_f"] - b["logmdd_f"]) <= 1e-9 * abs(b["logmdd_f"]), (a["logmdd_f"], b["logmdd_f"])
    for k in _KEYS:
        assert a[k] == c[k], (k, a[k], c[k])


@pytest.mark.parametrize("cfg", [
    dict(n=250_000, d=10, seed=3, kw=dict(use_fixed_schedule=False, tempering_target=0.97), history=False),
    dict(n=200_000, d=10, seed=5, kw=dict(use_fixed_schedule=True, n_phi=120), history=True),
    dict(n=150_004, d=6, seed=9, spec_args=[6], kw=dict(use_fixed_schedule=False, tempering_target=0.95, pause_at=9), history=False),
    dict(n=160_000, d=8, seed=4, spec_args=[8], kw=dict(use_fixed_schedule=True, n_phi=60, resampling_method="multinomial", threshold_ratio=0.8), history=False),
], ids=["adaptive_250000", "fixed_200000_history", "uneven_150004_pause", "fixed_160000_multinomial"])
def test_two_chunks_per_worker_keep_clouds_up_to_254000_particles_inside_segments(cfg):
    """One handle of 126 977 .. 253 952 particles (more 512-particle blocks than CUs) with α = 1, one block, one MH step and a cheap likelihood:
    every segment worker owns TWO chunks - one in registers, one parked in LDS, exchanged between the per-particle phases (stage3.hpp
    k3_segment<D, true, RIDE, 2>) - and publishes two rows per hand-over; the serial phases and the hand-overs are paid once (VERDICT r5 missing 4:
    the reference's loop has no size cliff, src/smc_main.jl:472-481).  Same bits as the same geometry's launches (SMCMI_ENGINE3=0); against
    engine 1 (SMCMI_ENGINE=1, what such a cloud ran on until round 5; sums in another order) the same stages and resample decisions and the
    log-MDD to rounding.  A stage that must resample does so inside the segment (k3_select_two: both chunks' rows, cum values and moment rows
    between the two hand-overs one chunk needs; SMCMI_SEG_SELECT=0: the segment leaves instead - the same bits)."""
    a = _run(cfg)[0]
    b = _run(cfg, {"SMCMI_ENGINE3": "0"})[0]
    c = _run(cfg, {"SMCMI_ENGINE": "1"})[0]
    nb2 = -(-(-(-cfg["n"] // 8)) // 512)
    assert a["n_segments"] >= 1 and b["n_segments"] == 0 and c["n_segments"] == 0
    off = _run(cfg, {"SMCMI_SEG_SELECT": "0"})[0]                        # the segments leave at the resample stages
    assert off["n_segments"] > a["n_segments"]
    for k in _KEYS:
        if k in off:
            assert a[k] == off[k], (k, a[k], off[k])
    assert a["segment_blocks"] == 8 * ((nb2 + 1) // 2) + 8 and a["segment_state"] == 1, (a["segment_blocks"], nb2)
    assert a["segment_stages"] >= (a["n_stages"] - 1) // 2
    for k in _KEYS:
        if k in b:
            assert a[k] == b[k], (k, a[k], b[k], a["stalls"], b["stalls"])
    assert (a["n_stages"], a["resamples"], a["resampled"]) == (c["n_stages"], c["resamples"], c["resampled"])
    assert abs(a["logmdd_f"] - c["logmdd_f"]) <= 1e-9 * abs(c["logmdd_f"]), (a["logmdd_f"], c["logmdd_f"])


def test_mh_bound_runs_of_that_size_stay_on_engine_1():
    """config 4's shape (three MH steps) at 200 000 particles: two chunks after each other at two wavefronts per SIMD are no faster than engine 1's
    mutation kernel at four (round 5: 30.6 against 29.6 ms) - run2.hpp two_chunk_run keeps such runs where they were."""
    cfg = dict(n=160_000, d=9, seed=6, spec="capm_spec", kw=dict(use_fixed_schedule=True, n_phi=40, n_mh_steps=3), history=False)
    a = _run(cfg)[0]
    assert a["n_segments"] == 0 and a["segment_blocks"] == 0


def test_a_segment_time_out_repeats_the_run_as_launches():
    """a hand-over inside a persistent segment that times out (the GPU shared after the residency self-test: not every block
    resident) voids the run with the cloud already overwritten.  A single-handle run keeps the cloud and the loop state it started from
    and repeats itself on engine 2's launches - here the time-out is forced by a bound no hand-over can meet (0.1 µs), on a fresh run
    (reps = 2: the second run of the handle starts on engine 2 right away) and on a run that pauses and continues."""
    for kw in (dict(use_fixed_schedule=False, tempering_target=0.95), dict(use_fixed_schedule=True, n_phi=50, pause_at=12)):
        cfg = dict(n=30_000, d=6, seed=21, spec_args=[6], kw=kw, reps=2)
        a = _run(cfg, {"SMCMI_SEG_TIMEOUT_MS": "0.0001"})
        b = _run(cfg, {"SMCMI_ENGINE3": "0"})
        for ra, rb in zip(a, b):
            assert ra["n_segments"] == 0                       # the result comes from the repeat: launches only
            for k in _KEYS:
                assert ra[k] == rb[k], (k, ra[k], rb[k])


def test_engine_1_two_level_totals_agree_with_engine_2():
    """Large single-handle clouds (engine 1): the prepare and stage-begin launches total their rows on two levels inside the launch
    (kernels.hpp PrepRed: reducer blocks + tickets).  The grouping changes the association of the sums, nothing else: against
    engine 2's canonical order (SMCMI_ENGINE=2; against one-block totals in rounds 4 and 5) the run has the same stages and
    resamples and its log-MDD agrees to rounding - at 200 000 particles (prepare reducers, random numbers drawn ahead by the same launch)
    and at 600 000 (the stage-begin reducers as well)."""
    for n in (200_000, 600_000):
        cfg = dict(n=n, d=10, seed=5, spec_args=[10], history=False, kw=dict(use_fixed_schedule=False, tempering_target=0.95))
        a = _run(cfg, {"SMCMI_ENGINE": "1"})[0]                                 # (200 000 particles of this model would run in two-chunk segments since round 6)
        c = _run(cfg, {"SMCMI_ENGINE": "2"})[0]
        assert a["n_segments"] == 0                                            # (engine 1: no segments)
        for other in (c,):
            assert (a["n_stages"], a["resamples"]) == (other["n_stages"], other["resamples"])
            assert abs(a["logmdd_f"] - other["logmdd_f"]) <= 1e-10 * abs(a["logmdd_f"]), (n, a["logmdd_f"], other["logmdd_f"])
