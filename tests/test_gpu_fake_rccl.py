"""The RCCL BRANCH of the sharded product driver with more than one rank.

`smcmi_comm_init` -> `smcmi_run_sharded` (csrc/sharded.hpp, run2.hpp) reach RCCL through ten dlopen()ed entry points; real RCCL needs a GPU
per rank and the project's boxes have one, so those lines - ncclGroupStart / Send / Recv / GroupEnd with the redistribution's counts and
displacements, the in-stream all-gathers and all-reduces, the peer mailbox's set-up THROUGH the communicator - had only ever run with one
rank.  tests/fake_rccl is a stand-in for librccl.so (same ten symbols, buffers moved through POSIX shared memory, mismatched collectives
reported instead of hanging) that SMCMI_RCCL_PATH points the library at: the driver's RCCL branch then runs as 2, 4 and EIGHT processes on
one GPU and must reproduce the single handle bit for bit - adaptive and fixed schedules, systematic (all-to-all-v of rows) and multinomial
(all-gather of the cloud) resampling, mailbox and all-gather hand-overs, small and large shards, and round 1's all-reduce driver (n_para > 16).
What replaces `@distributed` (src/smc_main.jl:169-170,472-476, src/resample.jl:33-35)."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests.test_gpu_multiproc import ROOT, _check, _single, _spawn

pytestmark = pytest.mark.gpu
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


@pytest.fixture(scope="module")
def fake_env():
    subprocess.check_call(["make", "-C", os.path.dirname(FAKE), "libfake_rccl.so"], stdout=subprocess.DEVNULL)
    return {"SMCMI_RCCL_PATH": FAKE}


def _logs(log_dir, world):
    out = []
    for r in range(world):
        with open(os.path.join(log_dir, "rank%d.log" % r)) as f:
            out.append([ln.split() for ln in f.read().splitlines()])
    return out


@pytest.mark.parametrize("world,inside", [(2, True), (2, False), (4, False), (8, True), (8, False)])
def test_rccl_branch_as_processes_on_one_gpu_reproduces_the_single_handle(world, inside, tmp_path, fake_env):
    """Adaptive schedule with resample stages, two runs per handle: the mailbox is mapped through ncclAllGather and agreed on through
    ncclAllReduce, the stages run inside sharded segments - at world = 8 the size the target node has (8 x 4 096 particles: one virtual shard
    per rank).  inside: the resample stages stay inside the segments as well (chunk sums, cum column and ancestors' rows through the mailbox
    allocation: csrc/stage3.hpp Sel3Args) - no collective call between the first and the last stage; else (SMCMI_SEG_SELECT=0) the segments
    leave at them and the rows travel through ncclSend / ncclRecv groups."""
    cfg = dict(n=32768, d=10, seed=7, comm="rccl", kw=dict(use_fixed_schedule=False, tempering_target=0.95), reps=2)
    want, want_cloud = _single(cfg)
    assert want["resamples"] >= 3
    log_dir = tmp_path / "log"
    log_dir.mkdir()
    extra = dict(fake_env, SMCMI_FAKE_RCCL_LOG=str(log_dir))
    if not inside:
        extra["SMCMI_SEG_SELECT"] = "0"
    runs, cloud = _spawn(world, cfg, tmp_path, env_extra=extra)
    _check(runs, cloud, want, want_cloud, expect_mailbox=True)
    # the call sequence: the same collectives with the same counts in the same order on every rank; every group's sends are received
    logs = _logs(str(log_dir), world)
    seq = [[(ln[0], ln[1]) for ln in lg if ln[0] in ("init", "allreduce", "allgather")] for lg in logs]
    assert all(s == seq[0] for s in seq), "ranks posted different collective sequences"
    groups = [[ln for ln in lg if ln[0] == "group"] for lg in logs]
    assert all(len(g) == len(groups[0]) for g in groups)
    if inside:
        assert len(groups[0]) == 0                                       # no row exchange by collectives: the selection ran inside the segments
        for rank_runs in runs:
            for r in rank_runs:
                assert r["segments"] <= 4 and r["segment_stages"] >= want["n_stages"] - 4, r
        return
    assert len(groups[0]) >= 2 * want["resamples"]                       # one group per resample stage and run (reps = 2)
    for k in range(len(groups[0])):
        sent = sum(int(g[k][3].split("=")[1]) for g in groups)
        got = sum(int(g[k][4].split("=")[1]) for g in groups)
        assert sent == got, (k, sent, got)
    assert any(int(g[k][3].split("=")[1]) > 0 for g in groups for k in range(len(g)))      # rows did cross ranks


@pytest.mark.parametrize("method,env", [("systematic", {"SMCMI_MAILBOX": "0"}), ("multinomial", {"SMCMI_MAILBOX": "0"}), ("systematic", {}), ("multinomial", {})])
def test_rccl_branch_fixed_schedule_both_resamplers_both_transports(method, env, tmp_path, fake_env):
    """Fixed schedule (the reference's default), two MH steps; hand-overs as ncclAllGather (SMCMI_MAILBOX=0) or through the mailbox; rows by
    send / recv groups (systematic) or by all-gathers of weights and clouds (multinomial) - with the mailbox the stages run as sharded segments
    and either resampler selects inside them.  Four ranks."""
    cfg = dict(n=16384, d=10, seed=11, comm="rccl", kw=dict(use_fixed_schedule=True, n_phi=40, n_mh_steps=2, resampling_method=method))
    want, want_cloud = _single(cfg)
    assert want["resamples"] >= 1
    runs, cloud = _spawn(4, cfg, tmp_path, env_extra=dict(fake_env, **env))
    _check(runs, cloud, want, want_cloud, expect_mailbox=not env)


def test_rccl_branch_large_shards(tmp_path, fake_env):
    """Two ranks of 200 000 particles (engine 2's large-shard stage, helper blocks polling the IPC-mapped mailbox), rows of resample stages
    through send / recv groups of ~1e5 rows x 16 columns."""
    cfg = dict(n=400000, d=10, seed=5, comm="rccl", kw=dict(use_fixed_schedule=False, tempering_target=0.95), reps=1)
    want, want_cloud = _single(cfg)
    assert want["resamples"] >= 2
    runs, cloud = _spawn(2, cfg, tmp_path, env_extra=fake_env)
    _check(runs, cloud, want, want_cloud, expect_mailbox=True)


def test_rccl_branch_stalls_pause_and_continue(tmp_path, fake_env):
    """Every resample stage arrives without its selection kernels (SMCMI_NO_SELECT_PREDICT=2), stalls on all ranks and is resumed behind an
    in-stream barrier (an ncclAllReduce of one double); a pause at a save point and a continuation.  Four ranks, two parameter blocks."""
    cfg = dict(n=16384, d=4, seed=3, spec_args=[4], comm="rccl", kw=dict(use_fixed_schedule=False, tempering_target=0.9, n_blocks=2, pause_at=7))
    want, want_cloud = _single(cfg, env_extra={"SMCMI_NO_SELECT_PREDICT": "2"})
    runs, cloud = _spawn(4, cfg, tmp_path, env_extra=dict(fake_env, SMCMI_NO_SELECT_PREDICT="2"))
    _check(runs, cloud, want, want_cloud, expect_mailbox=True)
    assert all(r["stalls"][1] >= 1 for rr in runs for r in rr)


def test_rccl_branch_of_the_all_reduce_driver_beyond_16_parameters(tmp_path, fake_env):
    """n_para > 16: round 1's sharded driver (sharded.hpp run_sharded_impl: engine 1's kernels between ncclAllReduce calls, the cloud's rows by
    send / recv on resample stages).  Two ranks against one handle on engine 1: same stages and resample decisions, log-MDD and cloud to
    rounding (the all-reduce adds the shards' partial sums in another order than one handle's blocks)."""
    cfg = dict(n=16384, d=20, seed=9, spec_args=[20], comm="rccl", full_records=True, kw=dict(use_fixed_schedule=True, n_phi=50, n_blocks=2))
    want, want_cloud = _single(cfg, env_engine2=False)
    runs, cloud = _spawn(2, cfg, tmp_path, env_extra=fake_env)
    for rr in runs:
        for r in rr:
            assert (r["n_stages"], r["resamples"]) == (want["n_stages"], want["resamples"])
            assert float.fromhex(r["logmdd"]) == pytest.approx(float.fromhex(want["logmdd"]), rel=1e-9)
    np.testing.assert_allclose(cloud, want_cloud, rtol=1e-7, atol=1e-9)


def test_mismatched_collectives_are_reported_not_hung(tmp_path, fake_env):
    """The stand-in's own contract: ranks that post different collectives get ncclInvalidUsage back (the driver turns it into SMCMI_ERR_HIP with
    the library's message) instead of the hang real RCCL would leave."""
    code = r'''
import os, sys, json
sys.path.insert(0, %r)
rank = int(sys.argv[1])
import ctypes as C
L = C.CDLL(%r)
class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
uid = Uid()
if rank == 0:
    assert L.ncclGetUniqueId(C.byref(uid)) == 0
    open(sys.argv[2], "wb").write(bytes(uid.b).ljust(128, b"\0"))
else:
    import time
    while not os.path.exists(sys.argv[2]) or os.path.getsize(sys.argv[2]) < 128: time.sleep(0.01)
    raw = open(sys.argv[2], "rb").read()
    C.memmove(C.byref(uid), raw, 128)
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")            # the runtime the stand-in itself links (torch ships another copy: its pointers would be foreign)
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
def dev(vals):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), 8 * len(vals)) == 0
    a = (C.c_double * len(vals))(*vals)
    assert hip.hipMemcpy(p, a, 8 * len(vals), 1) == 0
    return p
def host(p, n):
    a = (C.c_double * n)()
    assert hip.hipMemcpy(a, p, 8 * n, 2) == 0
    return list(a)
comm = C.c_void_p()
L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
assert L.ncclCommInitRank(C.byref(comm), 2, uid, rank) == 0
x, y = dev([1.0 + rank] * 8), dev([0.0] * 16)
L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
assert L.ncclAllGather(x, y, 8, 8, comm, None) == 0
assert host(y, 16) == [1.0] * 8 + [2.0] * 8
assert L.ncclAllReduce(x, x, 8, 8, 0, comm, None) == 0
assert host(x, 8) == [3.0] * 8
rc = L.ncclAllReduce(x, x, 8, 8, 0, comm, None) if rank == 0 else L.ncclAllGather(x, y, 8, 8, comm, None)
L.ncclGetErrorString.restype = C.c_char_p
print("RESULT " + json.dumps(dict(rc=rc, msg=L.ncclGetErrorString(rc).decode())))
''' % (ROOT, FAKE)
    uid_file = str(tmp_path / "uid")
    env = dict(os.environ, SMCMI_FAKE_RCCL_TIMEOUT_S="20")
    import sys
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), uid_file], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(2)]
    outs = []
    for r, p in enumerate(procs):
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, "rank %d: %s" % (r, se[-2000:])
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]))
    assert all(o["rc"] != 0 for o in outs), outs
    assert any("posted" in o["msg"] or "failed" in o["msg"] for o in outs), outs
