"""N > 1 path on CPU: tests.shard_orchestrator.ShardedSMC under torch.distributed/gloo with world_size 2.

The per-shard compute is the oracle-backed engine of tests/dist_helpers.py; what is under test is the orchestration
(collectives, replicated scalar logic, global-id RNG, resample exchange): a 2-rank sharded run must reproduce the
single-process oracle loop on the same seed.
"""
import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import dist_helpers, models


def _run_gloo(spec_name, n_parts, seed, kw, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (hash((spec_name, n_parts, seed)) % 2000)
    procs = [ctx.Process(target=dist_helpers.gloo_worker, args=(r, world, port, spec_name, n_parts, seed, kw, ret)) for r in range(world)]
    for p in procs:
        p.start()
    out = ret.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
    assert "error" not in out, out.get("error")
    return out


def _oracle(spec_name, n_parts, seed, kw):
    from oracle import oracle as orc

    spec = getattr(models, spec_name)()
    m = models.oracle_model(spec)
    P0 = orc.initial_draw(m, n_parts, seed=seed)
    okw = dict(kw)
    okw.pop("phi_rtol", None)
    return orc.smc_run(m, P0, seed=seed, n_threads=2, history=False, max_stages=2000, **okw)


@pytest.mark.parametrize("kw", [dict(use_fixed_schedule=False, tempering_target=0.95),
                                dict(use_fixed_schedule=True, n_phi=40, n_blocks=2, n_mh_steps=2, alpha=0.9,
                                     resampling_method="multinomial")])
def test_sharded_gloo_world2_matches_single_process(kw):
    n, seed = 3000, 21
    got = _run_gloo("regression_spec", n, seed, kw)
    want = _oracle("regression_spec", n, seed, kw)
    assert got["n_stages"] == want["n_stages"]
    assert got["resamples"] == want["resamples"]
    np.testing.assert_allclose(got["schedule"], want["schedule"], rtol=1e-9)
    np.testing.assert_allclose(got["ess"], want["ess"], rtol=1e-8)
    np.testing.assert_array_equal(got["resampled"], want["resampled"])
    assert got["logmdd"] == pytest.approx(want["logmdd"], abs=1e-8)
    np.testing.assert_allclose(got["cloud"], want["particles"], rtol=1e-7, atol=1e-9)


def test_host_solver_matches_oracle_bisection():
    """hostmath.PhiSolver (the algorithm the device solver runs) against the oracle's bit-level bisection."""
    from oracle import oracle as orc
    from smc_jl_amd.host import hostmath as hm

    rng = np.random.default_rng(2)
    n, d = 5000, 2
    P = np.zeros((n, d + 5), order="F")
    P[:, d] = -30 * rng.random(n)
    P[:, d + 4] = rng.random(n) + 0.5
    P[:, d + 4] *= n / P[:, d + 4].sum()
    sched = hm.schedule(300, 2.1)
    ess_now = orc.compute_ess(P[:, d], P[:, d + 4], 0.1, 0.1)

    def sums(c):
        v = P[:, d + 4][None, :] * np.exp((np.asarray(c)[:, None] - 0.1) * P[:, d][None, :])
        return v.sum(1), (v * v).sum(1)

    phi, j, phi_prop, passes = hm.PhiSolver(sched).solve(sums, 100, sched[98], 0.1, 0.97 * ess_now, ess_now)
    want = orc.solve_adaptive_phi(P, ess_now, sched, 100, sched[98], 0.1, 0.97, False)
    assert phi == pytest.approx(want[0], rel=1e-9) and j == want[2] and phi_prop == want[3]
    assert passes <= 7


def test_host_blocks_match_oracle():
    from oracle import oracle as orc
    from smc_jl_amd.host import hostmath as hm

    free = np.array([0, 2, 3, 5, 6, 7, 9], dtype=np.int32)
    for stage in (2, 17):
        for nb in (1, 2, 3):
            a = hm.generate_blocks(7, nb, free, 1234, stage)
            b = orc.generate_blocks(7, nb, free, 1234, stage)
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)


def _hostcomm_worker(rank, world, port, ret):
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smc_jl_amd.host.engine import torch_dist_host_comm

        allgather, alltoallv, barrier = torch_dist_host_comm()
        got = allgather(np.array([rank + 0.25, 10.0 * rank, -1.0]))
        # ragged all-to-all-v: rank r sends r + p + 1 doubles to peer p (nothing to itself)
        sends = [np.full(rank + p + 1, 100.0 * rank + p) if p != rank else np.zeros(0) for p in range(world)]
        recv_counts = [p + rank + 1 if p != rank else 0 for p in range(world)]
        recvs = alltoallv(sends, recv_counts)
        barrier()
        ret.put(dict(rank=rank, gathered=got.tolist(), recvs=[r.tolist() for r in recvs]))
    except Exception as ex:      # noqa: BLE001
        import traceback

        ret.put({"error": "rank %d: %s\n%s" % (rank, ex, traceback.format_exc())})
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_host_communicator_callables_under_gloo(world):
    """The transport `smcmi_comm_init_host` is given by the multi-process tests and by bench.py's host mode (engine.torch_dist_host_comm):
    rank order of the all-gather, ragged all-to-all-v, barrier - real processes, gloo, no GPU.  (The driver that calls them is HIP
    code: tests/test_gpu_multiproc.py runs it as 2 and 4 processes on one GPU.)"""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 31500 + world
    procs = [ctx.Process(target=_hostcomm_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [ret.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
    for o in outs:
        assert "error" not in o, o.get("error")
        want = []
        for r in range(world):
            want += [r + 0.25, 10.0 * r, -1.0]
        assert o["gathered"] == want
        me = o["rank"]
        for p in range(world):
            if p == me:
                assert o["recvs"][p] == []
            else:
                assert o["recvs"][p] == [100.0 * p + me] * (p + me + 1)
