"""Peer mailbox of sharded engine-2 runs (csrc/stage2.hpp Mailbox, run2.hpp): the transport that replaces the two all-gathers of a
stage when every rank could map every other rank's table.

A gpurun box has one GPU, RCCL refuses two ranks on one device, so the pieces are tested where they can be:
 * the protocol (tags, parities, counters rewound after a stalled stage, resample stages) inside one process: the in-process group
   driver posts into the other handles' tables directly (SMCMI_MAILBOX=1) and must reproduce the single-handle run bit for bit;
 * the IPC plumbing (export / import of the table handles, hipIpcOpenMemHandle, concurrent kernels of two processes exchanging
   rows through each other's tables) with two processes on the one GPU.
What no test here can see is the xGMI hop between two GPUs; the RCCL driver therefore runs 256 test exchanges on every rank before
it trusts the transport and keeps the all-gathers otherwise."""
import json
import os
import subprocess
import sys
import tempfile
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_IPC_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
from smc_jl_amd import Engine
from tests import models
rank, world, tmp = int(sys.argv[1]), 2, sys.argv[2]
e = Engine(8192, 6, seed=1, n_local=4096, gid0=rank * 4096, max_stages=50, store_history=False)
e.set_model(models.gauss_spec(6))
open(os.path.join(tmp, "h%%d.bin.tmp" %% rank), "wb").write(e.mailbox_export())
os.rename(os.path.join(tmp, "h%%d.bin.tmp" %% rank), os.path.join(tmp, "h%%d.bin" %% rank))
def wait(name, limit=400.0):
    t0 = time.time()
    while not os.path.exists(os.path.join(tmp, name)):
        if time.time() - t0 > limit: raise SystemExit("timed out waiting for " + name)
        time.sleep(0.01)
hs = []
for r in range(world):
    wait("h%%d.bin" %% r)
    hs.append(open(os.path.join(tmp, "h%%d.bin" %% r), "rb").read())
e.mailbox_import(rank, world, hs)
open(os.path.join(tmp, "ready%%d" %% rank), "w").write("1")
for r in range(world): wait("ready%%d" %% r)
errs = e.mailbox_selftest(rank, world, rounds=200)
print("ERRS %%d" %% errs, flush=True)
open(os.path.join(tmp, "done%%d" %% rank), "w").write("1")
for r in range(world): wait("done%%d" %% r)          # keep the table mapped until the peer has finished with it
e.close()
'''


def test_two_processes_exchange_rows_through_ipc_mapped_tables():
    with tempfile.TemporaryDirectory() as tmp:
        procs = [subprocess.Popen([sys.executable, "-c", _IPC_WORKER % dict(root=ROOT), str(r), tmp], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                  text=True, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(2)]
        outs = []
        for p in procs:
            try:
                o, e = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
        assert "ERRS 0" in o, (o, e[-1000:])


_GROUP_WORKER = r'''
import json, sys, hashlib
import numpy as np
sys.path.insert(0, %(root)r)
from smc_jl_amd import Engine, run_group
from tests import models
n, d, world, kw = 40000, 6, %(world)d, %(kw)r
engs = []
for r in range(world):
    e = Engine(n, d, seed=13, max_stages=1500, store_history=False, n_local=n // world, gid0=r * (n // world))
    e.set_model(models.gauss_spec(d)); e.init_from_prior(); engs.append(e)
res = run_group(engs, **kw) if world > 1 else engs[0].run(**kw)
cloud = np.concatenate([e.download_cloud() for e in engs], axis=0)
print("RESULT " + json.dumps(dict(n_stages=res["n_stages"], resamples=res["resamples"], logmdd=float(res["logmdd"]).hex(),
                                  cloud=hashlib.sha256(np.ascontiguousarray(cloud).tobytes()).hexdigest(),
                                  stalls=[res.get("solver_stalls", 0), res.get("select_stalls", 0), res.get("spec_stalls", 0)],
                                  segments=res["n_segments"], segment_stages=res["segment_stages"])))
'''


def _group(world, kw, env):
    p = subprocess.run([sys.executable, "-c", _GROUP_WORKER % dict(root=ROOT, world=world, kw=kw)], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


@pytest.mark.parametrize("kw,extra", [
    (dict(use_fixed_schedule=False, tempering_target=0.95, n_blocks=2, alpha=0.9), {}),
    (dict(use_fixed_schedule=True, n_phi=60, n_mh_steps=2), {}),
    (dict(use_fixed_schedule=False, tempering_target=0.9, resampling_method="multinomial"), {}),
    # every stage's selection is left out on purpose: each resample stage stalls and is resumed (tags rewound, fresh tags posted)
    (dict(use_fixed_schedule=False, tempering_target=0.95), {"SMCMI_NO_SELECT_PREDICT": "2"}),
])
def test_mailbox_hand_overs_reproduce_the_single_handle_bits(kw, extra):
    ref = _group(1, kw, dict(SMCMI_ENGINE="2", **extra))
    for world in (2, 8):
        got = _group(world, kw, dict(SMCMI_MAILBOX="1", **extra))
        for key in ("n_stages", "resamples", "logmdd", "cloud"):
            assert got[key] == ref[key], (world, key, got, ref)
        # with the mailbox up the persistent segments span the handles (stage3.hpp Seg3Args::peers): most stages ran inside them
        # (eight handles of ONE process share the device's four hardware queues - a persistent launch could sit in front of the launch it
        # waits for -, so the in-process driver keeps to launches beyond two handles; one handle per process has no such limit:
        # tests/test_gpu_multiproc.py)
        if world == 2:
            assert got["segments"] >= 1 and got["segment_stages"] >= (got["n_stages"] - 1) // 2, got
            off = _group(world, kw, dict(SMCMI_MAILBOX="1", SMCMI_ENGINE3="2", **extra))       # ... and as launches: the same bits
            assert off["segments"] == 0
            for key in ("n_stages", "resamples", "logmdd", "cloud"):
                assert off[key] == ref[key], (world, key, off, ref)
        else:
            assert got["segments"] == 0
    if extra:
        assert got["stalls"][1] >= 2          # the stall path really ran


def test_a_segment_time_out_in_a_group_repeats_the_run_as_launches():
    """A hand-over inside a sharded segment that runs out (forced: a bound nothing meets) stops that handle's posts, so every handle's run
    ends in the time-out; the group driver keeps every handle's starting cloud and loop state and repeats the run on the launches - the
    caller sees the single-handle result, from launches only."""
    kw = dict(use_fixed_schedule=False, tempering_target=0.95)
    ref = _group(1, kw, dict(SMCMI_ENGINE="2"))
    got = _group(2, kw, dict(SMCMI_MAILBOX="1", SMCMI_SEG_TIMEOUT_MS="0.0001"))
    assert got["segments"] == 0
    for key in ("n_stages", "resamples", "logmdd", "cloud"):
        assert got[key] == ref[key], (key, got, ref)


def test_rccl_driver_sets_the_mailbox_up_through_the_communicator():
    """bench.py under torch.distributed.run with one rank: smcmi_run_sharded exchanges the table handle through ncclAllGather, runs
    the 256-round self-test, agrees on the verdict through ncclAllReduce and then hands every stage's sums over through the table
    (SMCMI_MAILBOX=2: also with a single rank) - same stages, resamples and log-MDD bits as with the all-gather path."""
    out = []
    for mb in ("0", "2"):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SMCMI_FORCE_SHARDED="1", SMCMI_MAILBOX=mb)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29655", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--nparts", "40000",
               "--no-cpu", "--no-history"]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        out.append((d["n_stages"], d["resamples"], float(d["logmdd_gpu"]).hex()))
    assert out[0] == out[1]
