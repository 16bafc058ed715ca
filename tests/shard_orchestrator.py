"""ShardedSMC: TEST DOUBLE of the sharded loop - the stage sequence of csrc/sharded.hpp driven from Python over the shard-level C calls
(smcmi_shard_*), with the collectives as injectable callables.  The product path is smcmi_run_sharded (Engine.run_sharded; host closures
included since round 4); this file stays under tests/ because it lets the N > 1 logic run on CPU (gloo, world_size 2, an oracle-backed
engine: tests/test_distributed_cpu.py) and cross-checks the shard-level entry points on the GPU (tests/test_gpu_sharded.py).

Replaces the reference's only parallel mode - `@distributed` over particles with the whole cloud serialised to every
worker each stage (src/smc_main.jl:169-170, 472-476) - by resident shards plus a handful of tiny collectives:

  per ESS pass of the adaptive-ϕ solve : all-reduce of 2K doubles (Σv, Σv² per candidate)
  correction                          : all-reduce of (ΣW̃, ΣW̃²)  -> ESS, log-MDD increment, resample decision
  selection (ESS < threshold only)     : all-gather of the weights and of the cloud columns; every rank forms the same
                                         global cumulative sum and gathers the ancestors of ITS slots
  moments                              : all-reduce of 1 + d + d(d+1)/2 doubles
  mutation                             : all-reduce of Σ accept

Backend: torch.distributed ("nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).  Scalar decisions (ϕ solver,
c-adaptation, blocks, Cholesky inputs) are replicated on every rank from identical all-reduced numbers; per-particle RNG
uses global particle ids, so results do not depend on the shard count beyond floating-point summation order.
The per-shard compute is an `engine` object: smc_jl_amd.Engine (HIP, product) - the CPU tests inject an oracle-backed
engine with the same methods to exercise this file under world_size 2 without a GPU.
"""
import math
import time

import numpy as np

from smc_jl_amd.host import hostmath as hm


class TorchComm:
    """Collectives over torch.distributed (or a no-op when not initialised / world size 1)."""

    def __init__(self, device="cuda"):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        import os

        force = os.environ.get("SMCMI_FORCE_SHARDED") == "1"
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.device = device

    def all_reduce(self, x):
        x = np.asarray(x, dtype=np.float64)
        if not self.on:
            return x
        t = self.torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def all_gather(self, shard):
        """shard: torch tensor [..., n_local] -> [..., n_local * world] concatenated along the last axis in rank order."""
        if not self.on:
            return shard.clone()
        torch = self.torch
        lead = shard.shape[:-1]
        n = shard.shape[-1]
        flat = shard.reshape(-1, n)
        out = torch.empty((flat.shape[0], n * self.world), dtype=shard.dtype, device=shard.device)
        buf = torch.empty(self.world * n, dtype=shard.dtype, device=shard.device)
        for r in range(flat.shape[0]):
            self.dist.all_gather_into_tensor(buf, flat[r].contiguous())
            out[r] = buf
        return out.reshape(*lead, n * self.world)


class ShardedSMC:
    """`loglikelihood` / `old_loglikelihood`: host closures theta (m, d) -> (m,) log-likelihoods (the reference's
    `loglikelihood::Function` with `parallel = true`: every worker scores the particles it holds, src/smc_main.jl:472-476) - the
    spec's likelihood entries are then ("host_callback", [], None, None).  The shard's mutation runs as propose -> closure ->
    accept through the C ABI's split calls (smcmi_propose / smcmi_accept); nothing else of the loop changes."""

    def __init__(self, spec, n_parts, seed=0, device=0, max_stages=1500, store_history=False, engine=None, comm=None,
                 loglikelihood=None, old_loglikelihood=None):
        self.comm = comm if comm is not None else TorchComm("cuda" if engine is None else engine.tensor_device)
        world, rank = self.comm.world, self.comm.rank
        if n_parts % world:
            raise ValueError("n_parts must be divisible by the number of ranks")
        self.n_parts, self.n_local, self.gid0 = n_parts, n_parts // world, rank * (n_parts // world)
        self.d = len(spec["priors"])
        self.seed, self.max_stages = seed, max_stages
        if engine is None:
            from smc_jl_amd.host.engine import Engine

            engine = Engine(n_parts, self.d, seed=seed, device=device, max_stages=max_stages, store_history=store_history,
                            n_local=self.n_local, gid0=self.gid0)
            engine.set_model(spec)
        self.e = engine
        self.lik_fn, self.old_lik_fn = loglikelihood, old_loglikelihood
        if loglikelihood is not None:
            engine.set_likelihood_callback(loglikelihood, which=0)          # (the initial draw scores through it)
            if old_loglikelihood is not None:
                engine.set_likelihood_callback(old_loglikelihood, which=1)
        fixed = np.asarray(spec.get("fixed") or [0] * self.d)
        self.free_inds = np.flatnonzero(fixed == 0).astype(np.int32)
        self._snap = None

    # ---- cloud -----------------------------------------------------------------------------------------
    def init_from_prior(self):
        self.e.init_from_prior()

    def snapshot(self):
        self._snap = self.e.cloud_tensor().clone()

    def restore(self):
        t = self.e.cloud_tensor()
        t.copy_(self._snap)
        if getattr(t, "is_cuda", False):          # torch's stream wrote the cloud; the handle's non-blocking stream would not wait for it
            import torch

            torch.cuda.current_stream(t.device).synchronize()

    def download_cloud(self):
        return self.e.download_cloud()

    # ---- the loop (src/smc_main.jl:377-508) ---------------------------------------------------------------
    def run(self, n_blocks=1, n_mh_steps=1, lam=2.1, n_phi=300, resampling_method="systematic", threshold_ratio=0.5, c=0.5,
            alpha=1.0, target=0.25, use_fixed_schedule=True, tempering_target=0.97, prior_weight=0.0, log_prob_old_data=0.0,
            phi_rtol=1e-12):
        e, comm, N, d = self.e, self.comm, float(self.n_parts), self.d
        nf = len(self.free_inds)
        sched = hm.schedule(n_phi, lam)
        solver = hm.PhiSolver(sched, phi_rtol)
        i, j, phi_n, phi_prop = 1, 2, 0.0, 0.0
        resampled_last, threshold = False, threshold_ratio * N
        accept, logz, resamples = target, 0.0, 0
        ess_hist, phi_hist, c_hist, acc_hist, rs_hist = [N], [0.0], [c], [target], [0]
        shift = np.zeros(d)
        solver_passes = 0
        t0 = time.perf_counter()
        while phi_n < 1.0:
            i += 1
            if i > self.max_stages:
                raise RuntimeError("max_stages exceeded")
            phi_prev = phi_n
            if use_fixed_schedule:
                phi_n = float(sched[i - 1])
            else:
                if resampled_last:
                    ess_bar, ess_now, resampled_last = tempering_target * N, N, False          # helpers.jl:14-20
                else:
                    ess_bar, ess_now = tempering_target * ess_hist[-1], ess_hist[-1]

                def ess_sums(cands):
                    s1, s2 = e.shard_ess_sums(cands, phi_prev)
                    tot = comm.all_reduce(np.concatenate([s1, s2]))
                    return tot[:len(cands)], tot[len(cands):]

                phi_n, j, phi_prop, np_ = solver.solve(ess_sums, j, phi_prop, phi_prev, ess_bar, ess_now)
                solver_passes += np_
            tot = comm.all_reduce(e.shard_correct(phi_n, phi_prev, prior_weight, log_prob_old_data, i - 1))
            s1, s2 = float(tot[0]), float(tot[1])
            ess = s1 * s1 / s2
            if math.isnan(ess):
                raise FloatingPointError("No particles have non-zero weight (ESS is NaN)")   # check_nan_ess
            logz += math.log(s1 / N)
            resampled = ess < threshold
            if resampled:                                                                      # smc_main.jl:435-446
                full_cloud = comm.all_gather(e.cloud_tensor())          # [R, N]
                e.shard_resample(full_cloud[self.d + 4], full_cloud, resampling_method, i)
                resamples += 1
                resampled_last = True
            c = hm.update_c(c, accept, target)                                                  # :453-455
            mom = comm.all_reduce(e.shard_normalize_moments(s1, resampled, shift, i - 1))
            mean, cov = hm.moments_from_totals(mom, shift, d)
            shift = mean.copy()
            fi = self.free_inds
            mu_f, S_f = mean[fi], (cov[np.ix_(fi, fi)] + cov[np.ix_(fi, fi)].T) / 2.0           # :462-465
            bf, ba, bp = hm.generate_blocks(nf, n_blocks, fi, self.seed, i)                      # :468-469
            if self.lik_fn is None:
                acc_local = e.shard_mutate(mu_f, S_f, bp, bf, phi_n, phi_prev, c, alpha, n_mh_steps, i)
            else:
                acc_local = self._mutate_with_closures(mu_f, S_f, bp, bf, phi_n, c, alpha, n_mh_steps, n_blocks, i)
            asum = comm.all_reduce([acc_local])
            accept = float(asum[0]) / N                                                          # :484
            ess_hist.append(ess); phi_hist.append(phi_n); c_hist.append(c); acc_hist.append(accept); rs_hist.append(int(resampled))
        secs = time.perf_counter() - t0
        return dict(n_stages=i, resamples=resamples, logmdd=logz, c=c, accept=accept, seconds=secs,
                    schedule=np.array(phi_hist), ess=np.array(ess_hist), c_hist=np.array(c_hist),
                    accept_hist=np.array(acc_hist), resampled=np.array(rs_hist, dtype=np.int32), solver_passes=solver_passes)

    def _mutate_with_closures(self, mu_f, S_f, bp, bf, phi_n, c, alpha, n_mh_steps, n_blocks, stage):
        """mutation! (src/mutation.jl:56-138) of this shard with the likelihoods on the host: per (MH step, block) the device proposes
        (mixture draw, proposal densities, bounds, prior), the closures score the proposals inside the bounds, the device decides.
        Returns the shard's Σ accept (update_acceptance_rate!, src/particle.jl:466-468)."""
        e = self.e
        nb = len(bp) - 1
        for step in range(n_mh_steps):
            for b in range(nb):
                prop, lpr, _ = e.propose(mu_f, S_f, bp, bf, b, step, c, alpha, stage)
                inside = np.isfinite(lpr)                                   # out of bounds => ParamBoundsError => -Inf everywhere
                ln = np.full(e.n, -np.inf)
                lo = None
                if inside.any():
                    ln[inside] = np.asarray(self.lik_fn(np.ascontiguousarray(prop[inside])), dtype=np.float64)
                if self.old_lik_fn is not None:
                    lo = np.full(e.n, -np.inf)
                    if inside.any():
                        lo[inside] = np.asarray(self.old_lik_fn(np.ascontiguousarray(prop[inside])), dtype=np.float64)
                e.accept(ln, lo, phi_n, b, step, nb, stage, last=(step == n_mh_steps - 1 and b == nb - 1))
        e.sync()                                                            # (the handle's stream is not torch's)
        return float(e.cloud_tensor()[self.d + 3].sum().item())

