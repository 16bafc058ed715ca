"""The sharded product driver as SEVERAL PROCESSES: what replaces `@distributed` (src/smc_main.jl:472-476,
src/resample.jl:33-35) is `smcmi_run_sharded` -> csrc/run2.hpp `run2_impl`; with RCCL it needs one GPU per rank, so until now it had
only ever executed with one rank or as in-process handles.  Here it runs as 2 and 4 processes that share the one GPU of the box, over
the host-mediated communicator (smcmi_comm_init_host; collectives by torch.distributed / gloo on host buffers), the per-stage sums
handed over through the peer mailbox mapped between the processes with hipIpcOpenMemHandle - and must reproduce the single handle
bit for bit: same ϕ schedule, ESS path, acceptance rates, log-MDD, cloud."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import models

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(world, cfg, tmp_path, env_extra=None, timeout=900):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    out = str(tmp_path)
    procs = [subprocess.Popen([sys.executable, "-m", "tests.mp_shard_worker", str(r), str(world), str(port), out, json.dumps(cfg)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    errs = []
    for r, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        if p.returncode != 0:
            errs.append("rank %d rc %d:\n%s" % (r, p.returncode, se[-3000:]))
    assert not errs, "\n".join(errs)
    runs = [json.load(open(os.path.join(out, "rank%d.json" % r))) for r in range(world)]
    cloud = np.concatenate([np.load(os.path.join(out, "cloud%d.npy" % r)) for r in range(world)], axis=0)
    return runs, cloud


def _single(cfg, env_engine2=True, env_extra=None):
    """The same population on ONE handle, in a fresh process (SMCMI_ENGINE=2: the engine every sharded run uses)."""
    code = r'''
import json, sys, hashlib
import numpy as np
sys.path.insert(0, %r)
from smc_jl_amd import Engine
from tests import models
cfg = json.loads(%r)
spec = getattr(models, cfg.get("spec", "gauss_spec"))(*cfg.get("spec_args", []))
e = Engine(cfg["n"], cfg["d"], seed=cfg["seed"], max_stages=cfg.get("max_stages", 1500), store_history=False)
e.set_model(spec); e.init_from_prior()
kw = dict(cfg["kw"]); stop = kw.pop("pause_at", 0)
if stop:          # (a continuation re-enters the solver without a prediction: pause the single handle at the same stage)
    r = e.run(stop_after_stage=stop, **kw); assert r["paused"]
    r = e.run(continue_run=True, **kw)
else:
    r = e.run(**kw)
rec = e.stage_records(r["n_stages"])
np.save(sys.argv[1], e.download_cloud())
print("RESULT " + json.dumps(dict(n_stages=r["n_stages"], resamples=r["resamples"], logmdd=float(r["logmdd"]).hex(),
      schedule=hashlib.sha256(np.ascontiguousarray(rec["schedule"]).tobytes()).hexdigest(),
      ess=hashlib.sha256(np.ascontiguousarray(rec["ess"]).tobytes()).hexdigest(),
      accept=hashlib.sha256(np.ascontiguousarray(rec["accept_hist"]).tobytes()).hexdigest())))
''' % (ROOT, json.dumps(cfg))
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "c.npy")
        env = dict(os.environ, SMCMI_ENGINE="2") if env_engine2 else dict(os.environ)
        env.update(env_extra or {})
        p = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
        return res, np.load(path)


def _check(runs, cloud, want, want_cloud, expect_mailbox):
    for rank_runs in runs:
        for r in rank_runs:
            for key in ("n_stages", "resamples", "logmdd", "schedule", "ess", "accept"):
                assert r[key] == want[key], (key, r[key], want[key])
            if expect_mailbox is not None:
                assert r["mailbox"] == expect_mailbox
    assert hashlib.sha256(np.ascontiguousarray(cloud).tobytes()).hexdigest() == hashlib.sha256(np.ascontiguousarray(want_cloud).tobytes()).hexdigest()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_processes_sharing_one_gpu_reproduce_the_single_handle(world, tmp_path):
    """Adaptive schedule (resample stages among the predicted ones), mailbox mapped across processes for the whole run."""
    cfg = dict(n=32768, d=10, seed=7, kw=dict(use_fixed_schedule=False, tempering_target=0.95), reps=2)
    want, want_cloud = _single(cfg)
    assert want["resamples"] >= 3
    runs, cloud = _spawn(world, cfg, tmp_path)
    _check(runs, cloud, want, want_cloud, expect_mailbox=True)
    # the resample stages stay inside the sharded segments (csrc/stage3.hpp Sel3Args: chunk sums, cum column and rows through the mailbox allocation)
    for rank_runs in runs:
        for r in rank_runs:
            assert r["segments"] <= 4 and r["segment_stages"] >= want["n_stages"] - 4, r


@pytest.mark.parametrize("world", [2, 8])
def test_fixed_schedule_segments_across_processes_take_one_hand_over_per_stage(world, tmp_path):
    """The reference's default schedule on sharded segments: every rank's segment kernel rides (stage n+1's correction row on stage n's
    mutation row, the totals tables in two copies inside the mailbox allocation, csrc/stage3.hpp K3_TPAR) - the bits of one handle, and of the
    same ranks with exact energy shifts switched on from a stage in the middle of the run (SMCMI_SHIFT_LAG=k: the fallback of an overflowing sum)."""
    cfg = dict(n=32768, d=10, seed=9, kw=dict(use_fixed_schedule=True, n_phi=120), reps=2)
    want, want_cloud = _single(cfg)
    assert want["resamples"] >= 3
    runs, cloud = _spawn(world, cfg, tmp_path)
    _check(runs, cloud, want, want_cloud, expect_mailbox=True)
    for rank_runs in runs:
        for r in rank_runs:
            assert r["segments"] >= 1 and r["segment_stages"] >= 100, r
            assert r["shift_fallback_stage"] == 0, r
    if world == 2:
        sub = tmp_path / "fb"
        sub.mkdir()
        want_fb, cloud_fb = _single(cfg, env_extra={"SMCMI_SHIFT_LAG": "40"})
        runs_fb, cloud2 = _spawn(world, cfg, sub, env_extra={"SMCMI_SHIFT_LAG": "40"})
        _check(runs_fb, cloud2, want_fb, cloud_fb, expect_mailbox=True)
        assert all(r["shift_fallback_stage"] == 40 for rr in runs_fb for r in rr), runs_fb


@pytest.mark.parametrize("n", [66002, 130046, 104092])
def test_sharded_segments_total_virtual_shards_of_more_than_64_rows(n, tmp_path):
    """2 x odd particles: two virtual shards of 65 / 127 rows - what a rank of an 8-GPU run of 260 000 .. 520 000 particles holds.  The
    gatherer of such a shard totals TWO canonical groups of rows (csrc/stage3.hpp gather_vshard; until round 6 it left the second one out:
    wrong totals, found by this configuration).  One rank with the mailbox forced on - the only way such a shard's workers fit one GPU.
    104 092 = 4 x odd: four shards of 51 rows - one handle cut its gather blocks into two tiles there until round 6 (run2.hpp make_geo2: nbg),
    i.e. summed the resampled cloud's moments in another order than the same cloud on several handles: last-bit differences."""
    cfg = dict(n=n, d=10, seed=7, kw=dict(use_fixed_schedule=False, tempering_target=0.95), reps=1)
    want, want_cloud = _single(cfg)
    runs, cloud = _spawn(1, cfg, tmp_path, env_extra={"SMCMI_MAILBOX": "2"})
    _check(runs, cloud, want, want_cloud, expect_mailbox=True)
    assert runs[0][0]["segments"] >= 1 and runs[0][0]["segment_stages"] >= want["n_stages"] // 2, runs[0][0]


def test_large_shards_across_processes(tmp_path):
    """Two processes of 200 000 particles each - shards beyond one 512-particle block per CU run engine 2's large-shard stage
    (csrc/stage2b.hpp): the helper block of every rank's K1 launch polls the IPC-mapped mailbox for BOTH ranks' correction totals, the helper
    block of the mutation launch for both ranks' mutation totals; only those two blocks ever wait, so ranks that share the one GPU cannot
    starve each other.  Same bits as one handle of 400 000."""
    cfg = dict(n=400000, d=10, seed=5, kw=dict(use_fixed_schedule=False, tempering_target=0.95), reps=1)
    want, want_cloud = _single(cfg)
    assert want["resamples"] >= 2
    runs, cloud = _spawn(2, cfg, tmp_path)
    _check(runs, cloud, want, want_cloud, expect_mailbox=True)


def test_stalled_and_resumed_stages_across_processes(tmp_path):
    """SMCMI_NO_SELECT_PREDICT=2: every resample stage arrives without its selection kernels, stalls on all ranks and is resumed by the
    hosts (fresh mailbox tags after a barrier); plus a pause at a save point and a continuation.  Same bits as one handle."""
    cfg = dict(n=16384, d=4, seed=3, spec_args=[4], kw=dict(use_fixed_schedule=False, tempering_target=0.9, n_blocks=2, pause_at=7))
    # (the switch also makes resample stages take the predicted ϕ_n instead of a certified one: the single handle runs under it too)
    want, want_cloud = _single(cfg, env_extra={"SMCMI_NO_SELECT_PREDICT": "2"})
    runs, cloud = _spawn(2, cfg, tmp_path, env_extra={"SMCMI_NO_SELECT_PREDICT": "2"})
    _check(runs, cloud, want, want_cloud, expect_mailbox=True)
    assert all(r["stalls"][1] >= 1 for rr in runs for r in rr)


@pytest.mark.parametrize("env", [{"SMCMI_MAILBOX": "0"}, {"SMCMI_MAILBOX": "0", "SMCMI_RESAMPLE_EXCHANGE": "allgather"}])
def test_all_gather_hand_overs_and_multinomial_across_processes(env, tmp_path):
    """The fall-back transport (every hand-over an all-gather through the host communicator), fixed schedule, multinomial resampling
    (rows by all-gather) and systematic (all-to-all-v through the communicator's alltoallv)."""
    method = "multinomial" if "SMCMI_RESAMPLE_EXCHANGE" in env else "systematic"
    cfg = dict(n=16384, d=10, seed=11, kw=dict(use_fixed_schedule=True, n_phi=40, n_mh_steps=2, resampling_method=method))
    want, want_cloud = _single(cfg)
    runs, cloud = _spawn(2, cfg, tmp_path, env_extra=env)
    _check(runs, cloud, want, want_cloud, expect_mailbox=False)


@pytest.mark.parametrize("world,comm", [(2, "host"), (8, "rccl_shared")])
def test_bench_py_with_several_ranks_on_one_gpu(world, comm):
    """bench.py's whole multi-rank path - rank set-up under torch.distributed.run, the pre-flight that checks the mailbox against the
    all-gathers bit for bit, the timed steps, the max-over-ranks time, the single-GPU reference, the JSON line - had never executed
    with more than one rank (one GPU per box).  SMCMI_BENCH_COMM=host lets the ranks share the GPU over the host-mediated communicator;
    =rccl_shared over the library's RCCL branch (SMCMI_RCCL_PATH -> tests/fake_rccl, the shared-memory stand-in) at EIGHT ranks, the size of
    the target node: the line's `config.preflight` says what every rank's pre-flight found."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SMCMI_BENCH_COMM=comm)
    if comm == "rccl_shared":
        fake = os.path.join(ROOT, "tests", "fake_rccl")
        subprocess.check_call(["make", "-C", fake, "libfake_rccl.so"], stdout=subprocess.DEVNULL)
        env["SMCMI_RCCL_PATH"] = os.path.join(fake, "libfake_rccl.so")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--nparts", "65536", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-history"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["value"] > 0 and d["steps"] == 2
    assert d["config"]["n_parts_total"] == 65536 and d["config"]["n_parts_per_gpu"] == 65536 // world
    assert "one GPU" in d["config"]["hand_over"]
    pf = d["config"]["preflight"]
    assert pf["ranks"] == world and pf["default_run_ok"] == world and pf["bits_equal"] and pf["mailbox_ok"] and pf["mailbox_ranks"] == world, pf
    assert pf["segment_ranks"] == world, pf                       # the stages ran inside segments that span the ranks
    assert d["speedup_vs_single_gpu"] > 0 and d["logmdd_abs_diff_vs_single_gpu"] == 0.0      # the same bits as the single handle (engine 3 there, engine 2 here)
    assert abs(d["logmdd_gpu"] - models.gauss_logmdd(10)) < 0.3


@pytest.mark.parametrize("mode", ["closure", "tempered"])
def test_host_closures_through_the_sharded_product_driver(mode, tmp_path):
    """VERDICT r3 missing 3: the reference's default use - smc(loglikelihood::Function, ...) with parallel = true, every worker scoring
    the particles it holds (src/smc_main.jl:118, 472-476) - through smcmi_run_sharded itself: two PROCESSES, each with its shard and its
    registered closure (propose kernel -> closure -> accept kernel per MH step x block), the collectives over the host communicator;
    against ONE handle running the same closure through smcmi_run's callback path.  Same stages and resample decisions, log-MDD to 1e-8,
    the same cloud (particle ids are global: a shard proposes exactly what the single handle proposes for its rows).  "tempered": an
    old-data closure beside the new one (generalized tempering, src/mutation.jl:96-106)."""
    cfg = dict(n=20000, d=5, seed=21, spec_args=[5], closure=mode, kw=dict(use_fixed_schedule=False, tempering_target=0.9, n_blocks=2, n_mh_steps=2, alpha=0.9))
    code = r'''
import json, sys
import numpy as np
sys.path.insert(0, %r)
from smc_jl_amd import Engine
from tests import models
cfg = json.loads(%r)
spec = models.gauss_spec(*cfg["spec_args"])
cl = models.gauss_closures(spec, tempered=cfg["closure"] == "tempered")
spec = dict(spec, lik=("host_callback", [], None, None), old_lik=("host_callback", [], None, None) if cfg["closure"] == "tempered" else None)
e = Engine(cfg["n"], cfg["d"], seed=cfg["seed"], max_stages=1500, store_history=False)
e.set_model(spec)
for which, fn in enumerate(cl["fns"]): e.set_likelihood_callback(fn, which=which)
e.init_from_prior()
if cfg["closure"] == "tempered": e.eval_cloud_callback(which=1, column=cfg["d"] + 2)
r = e.run(**cfg["kw"])
rec = e.stage_records(r["n_stages"])
np.save(sys.argv[1], e.download_cloud())
print("RESULT " + json.dumps(dict(n_stages=r["n_stages"], resamples=r["resamples"], logmdd=r["logmdd"], schedule=[float(x) for x in rec["schedule"]], calls=e.callback_stats()["calls"])))
''' % (ROOT, json.dumps(cfg))
    path = str(tmp_path / "single.npy")
    p = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    want = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    want_cloud = np.load(path)
    assert want["n_stages"] > 10 and want["resamples"] >= 1 and want["calls"] >= (want["n_stages"] - 1) * 4
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = str(tmp_path)
    wcfg = dict(cfg, full_records=True)
    procs = [subprocess.Popen([sys.executable, "-m", "tests.mp_shard_worker", str(r), "2", str(port), out, json.dumps(wcfg)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for r, pr in enumerate(procs):
        so, se = pr.communicate(timeout=900)
        assert pr.returncode == 0, "rank %d: %s" % (r, se[-3000:])
    runs = [json.load(open(os.path.join(out, "rank%d.json" % r)))[0] for r in range(2)]
    cloud = np.concatenate([np.load(os.path.join(out, "cloud%d.npy" % r)) for r in range(2)], axis=0)
    for r in runs:
        assert r["n_stages"] == want["n_stages"] and r["resamples"] == want["resamples"]
        np.testing.assert_allclose(r["schedule_values"], want["schedule"], rtol=1e-9)
        assert float.fromhex(r["logmdd"]) == pytest.approx(want["logmdd"], abs=1e-8)
    np.testing.assert_allclose(cloud, want_cloud, rtol=1e-7, atol=1e-9)
