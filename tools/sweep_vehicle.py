"""development: randomised configurations on the ONE-RANK VEHICLE of a sharded run - a communicator of one rank with the peer mailbox forced on
(SMCMI_MAILBOX=2): the system-scope hand-overs, the sharded segments and their in-place selection, on whatever cut the particle count gets
(8 / 4 / 2 virtual shards or one, up to 128 rows each) - against one handle on engine 2's launches (SMCMI_ENGINE=2): the bits must agree.
usage: python tools/sweep_vehicle.py <seed> <trials>"""
import json, os, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_multiproc import _spawn, _single, _check

# SWEEP_D=lo,hi: range of n_para (default 1..10: the register kernels and the segments; 11..16: the wide two-launch stage, at most 65 536 per shard)
DLO, DHI = [int(x) for x in os.environ.get("SWEEP_D", "1,10").split(",")]
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    d = int(rs.randint(DLO, DHI + 1))
    nb = int(rs.randint(1, min(d, 3) + 1))
    while ((d + nb - 1) // nb) * (nb - 1) >= d: nb -= 1
    kw = dict(n_blocks=nb, n_mh_steps=int(rs.randint(1, 3)), alpha=float(rs.choice([1.0, 0.9, 0.5])), use_fixed_schedule=bool(rs.randint(0, 2)),
              n_phi=int(rs.choice([30, 60])), tempering_target=float(rs.choice([0.9, 0.95])), resampling_method=str(rs.choice(["systematic", "multinomial"])),
              threshold_ratio=float(rs.choice([0.5, 0.8])))
    if rs.randint(0, 3) == 0: kw["pause_at"] = int(rs.randint(3, 25))      # an intermediate save point, continued in place (both sides alike)
    if not kw["use_fixed_schedule"] and kw.get("pause_at", 0) > 8: kw["pause_at"] = 3 + kw["pause_at"] % 6      # (a short adaptive run must still reach it)
    # the cut: V = 8 (n = 8 k, up to 31 rows per shard), 4 (n = 4 odd, up to 63), 2 (n = 2 odd, up to 127), 1 (n odd, up to 128 rows)
    V = int(rs.choice([8, 4, 2, 1]))
    rows = int(rs.randint(1, {8: 31, 4: 63, 2: 127, 1: 128}[V] + 1))
    per = (rows - 1) * 512 + int(rs.randint(1, 513))
    if V > 1 and per % 2 == 0: per -= 1                      # an odd shard: no larger cut divides n
    per = max(per, 1)
    if V == 1 and per % 2 == 0: per = max(1, per - 1)
    n = V * per
    if n < 64: n = 64 * V + V * (per % 2 == 0)
    cfg = dict(n=n, d=d, seed=int(rs.randint(1, 1000)), spec_args=[d], kw=kw, reps=1, max_stages=400)
    try:
        want, want_cloud = _single(cfg)
        td = pathlib.Path(tempfile.mkdtemp())
        runs, cloud = _spawn(1, cfg, td, env_extra={"SMCMI_MAILBOX": "2"})
        if d > 16:     # round 1's all-reduce driver adds the shards' partial sums in another order than one handle's blocks: decisions equal, values to rounding
            r0 = runs[0][0]
            assert (r0["n_stages"], r0["resamples"]) == (want["n_stages"], want["resamples"]), (r0["n_stages"], r0["resamples"], want["n_stages"], want["resamples"])
            assert abs(float.fromhex(r0["logmdd"]) - float.fromhex(want["logmdd"])) <= 1e-9 * abs(float.fromhex(want["logmdd"])), (r0["logmdd"], want["logmdd"])
            np.testing.assert_allclose(cloud, want_cloud, rtol=1e-6, atol=1e-8)
        else:
            _check(runs, cloud, want, want_cloud, expect_mailbox=True if d <= 10 else None)        # (n_para > 10: whatever transport the wide stage takes)
        r = runs[0][0]
        print(json.dumps(dict(trial=trial, ok=True, V=V, n=n, d=d, **kw, stages=r["n_stages"], resamples=r["resamples"], segments=r["segments"], segment_stages=r["segment_stages"])), flush=True)
    except AssertionError as ex:
        bad += 1
        print(json.dumps(dict(trial=trial, ok=False, V=V, n=n, d=d, **kw, err=str(ex)[-600:])), flush=True)
print("bad", bad)
