"""Test-only helpers for the sharded path: an oracle-backed shard engine (CPU) and an in-process thread communicator.

OracleShardEngine implements the engine protocol of tests.shard_orchestrator.ShardedSMC with numpy + the CPU oracle,
so the orchestration (collectives, replicated scalar logic, resample exchange) runs under gloo without a GPU.
It lives under tests/ because only tests may touch oracle/.
"""
import threading

import numpy as np


class OracleShardEngine:
    tensor_device = "cpu"

    def __init__(self, spec, n_parts, n_local, gid0, seed):
        from oracle import oracle as orc
        from tests import models

        self.orc, self.model = orc, models.oracle_model(spec)
        self.n_parts, self.n, self.gid0, self.seed = n_parts, n_local, gid0, seed
        self.d = len(spec["priors"])
        self.R = self.d + 5
        self.P = np.zeros((self.n, self.R), order="F")

    def init_from_prior(self):
        self.P = self.orc.initial_draw(self.model, self.n, seed=self.seed, pid0=self.gid0)

    def download_cloud(self):
        return self.P.copy(order="F")

    def cloud_tensor(self):
        import torch

        return torch.from_numpy(self.P.T)       # [R, n] view sharing memory with the Fortran-ordered cloud

    def shard_ess_sums(self, phis, phi_prev):
        d, P = self.d, self.P
        s1, s2 = [], []
        for ph in np.atleast_1d(phis):
            v = P[:, d + 4] * np.exp((phi_prev - ph) * P[:, d + 2] + (ph - phi_prev) * P[:, d])
            s1.append(v.sum()); s2.append((v * v).sum())
        return np.array(s1), np.array(s2)

    def shard_correct(self, phi_n, phi_prev, pw, logp_old, stage_col):
        d, P = self.d, self.P
        if pw == 0.0:
            inc = np.exp((phi_prev - phi_n) * P[:, d + 2] + (phi_n - phi_prev) * P[:, d])
        elif pw == 1.0:
            inc = np.exp((phi_n - phi_prev) * P[:, d])
        else:
            inc = np.exp((phi_prev - phi_n) * np.log(np.exp(P[:, d + 2] - logp_old + np.log(1 - pw)) + pw) + (phi_n - phi_prev) * P[:, d])
        P[:, d + 4] *= inc
        return np.array([P[:, d + 4].sum(), (P[:, d + 4] ** 2).sum()])

    def shard_normalize_moments(self, sum_unnorm, resampled, shift, stage_col):
        d, P = self.d, self.P
        if not resampled:
            P[:, d + 4] = (P[:, d + 4] * self.n_parts) / sum_unnorm
        w = P[:, d + 4]
        X = np.concatenate([np.ones((self.n, 1)), P[:, :d] - np.asarray(shift)], axis=1)
        out = []
        for a in range(d + 1):
            for b in range(a, d + 1):
                out.append(np.sum(w * X[:, a] * X[:, b]))
        return np.array(out)

    def shard_mutate(self, mu_f, S_f, bp, bf, phi_n, phi_prev, c, alpha, n_mh, stage):
        ba = self.model.free_inds[np.asarray(bf)]
        self.P = self.orc.mutate_cloud(self.model, self.P, mu_f, S_f, bf, ba, bp, phi_n, phi_prev, c, alpha, n_mh, self.seed,
                                       stage, pid0=self.gid0)
        return float(self.P[:, self.d + 3].sum())

    def shard_resample(self, full_weights, full_cloud, method, stage):
        fw = full_weights.numpy()
        idx = self.orc.resample(fw, method, seed=self.seed, stage=stage)[self.gid0:self.gid0 + self.n]
        rows = full_cloud.numpy()[:, idx].T          # [n, R]
        self.P = np.asfortranarray(rows.copy())
        self.P[:, self.d + 4] = 1.0
        return idx


class ThreadComm:
    """All-reduce / all-gather between threads of one process (simulates ranks for single-GPU tests)."""

    class _Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world

    def all_reduce(self, x):
        x = np.asarray(x, dtype=np.float64)
        self.s.slots[self.rank] = x
        self.s.barrier.wait()
        tot = np.zeros_like(x)
        for r in range(self.world):          # fixed rank order on every rank
            tot = tot + self.s.slots[r]
        self.s.barrier.wait()
        return tot

    def all_gather(self, shard):
        import torch

        self.s.slots[self.rank] = shard.clone()
        self.s.barrier.wait()
        out = torch.cat([self.s.slots[r] for r in range(self.world)], dim=-1)
        self.s.barrier.wait()
        return out


def gloo_worker(rank, world, port, spec_name, n_parts, seed, kw, ret):
    """Entry point of one gloo rank (torch.multiprocessing.spawn)."""
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.shard_orchestrator import ShardedSMC, TorchComm
        from tests import models

        spec = getattr(models, spec_name)()
        comm = TorchComm("cpu")
        n_local = n_parts // world
        eng = OracleShardEngine(spec, n_parts, n_local, rank * n_local, seed)
        sm = ShardedSMC(spec, n_parts, seed=seed, engine=eng, comm=comm, max_stages=2000)
        sm.init_from_prior()
        r = sm.run(**kw)
        import torch

        full = comm.all_gather(eng.cloud_tensor().clone())
        if rank == 0:
            r["cloud"] = np.asfortranarray(full.numpy().T)
            ret.put(r)
    except Exception as ex:          # surface the failure instead of letting the parent wait for its timeout
        import traceback

        ret.put({"error": "rank %d: %s\n%s" % (rank, ex, traceback.format_exc())})
        raise
    finally:
        dist.destroy_process_group()
