#!/usr/bin/env python
"""bench.py - particle-stages/sec of the SMC correction/selection/mutation loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete pass of the hot path over one batch: a full smc() run (all tempering stages) of
BASELINE config 2 - 10-dim isotropic-Gaussian log-likelihood, n_parts = 100k per GPU, adaptive ϕ
(tempering_target 0.97, n_Φ = 300, λ = 2.1), systematic resampling, 1 block, 1 MH step - on synthetic
prior draws made on the device at the start of every step (`initial_draw!`; same seed => the same cloud in every step,
nothing crosses PCIe).  value = n_parts * (n_stages - 1) * K / wall: the reference's metric (stage bracket
src/smc_main.jl:378,489-490) with the device-side initial draw inside the wall time as well; file I/O excluded.  For N > 1 the driver
launches one rank per GPU (torch.distributed, RCCL); particles are sharded (weak scaling: 100k per GPU) with
small all-reduces per stage and an exchange on resample stages.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D = 10
N_PER_GPU = 100_000
RUN_KW = dict(use_fixed_schedule=False, tempering_target=0.97, n_phi=300, lam=2.1, resampling_method="systematic",
              n_blocks=1, n_mh_steps=1, alpha=1.0, c=0.5, target=0.25, threshold_ratio=0.5)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def mutate_bytes_per_particle(d):
    # mutate reads θ(d), ℓ, π, ℓ_old and writes θ(d), ℓ, π, ℓ_old, accept = 16 d + 56 bytes (FP64)
    return 16 * d + 56


def main():
    # A collective that never completes (a rank died, a transport misbehaves) must not hold the GPUs until somebody's outer limit
    # fires: dump the Python stacks and exit after SMCMI_BENCH_WATCHDOG seconds (default 15 min; the default run takes ~1 min).
    import faulthandler

    faulthandler.dump_traceback_later(int(os.environ.get("SMCMI_BENCH_WATCHDOG", "900")), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nparts", type=int, default=N_PER_GPU, help="particles per GPU")
    ap.add_argument("--mode", default="direct", choices=["direct", "graph"], help="stage launch mode")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU-oracle baseline leg")
    ap.add_argument("--no-history", action="store_true")
    ap.add_argument("--workload", default="gauss10", choices=["gauss10", "capm", "kalman"],
                    help="gauss10 = BASELINE config 2 (the bench line); capm = config 4 (examples/capm_model, 3 MH steps, fixed schedule); kalman = config 5 (13-parameter state-space model, Kalman-filter likelihood, old + new data)")
    ap.add_argument("--solver-passes", type=int, default=0)
    ap.add_argument("--sync-every", type=int, default=0)
    ap.add_argument("--phi-rtol", type=float, default=0.0, help="adaptive-phi root tolerance (0 = library default)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from tests import models

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    torch.cuda.set_device(local_rank)
    dist = None
    force_sharded = os.environ.get("SMCMI_FORCE_SHARDED") == "1"
    if world > 1 or (force_sharded and "RANK" in os.environ):
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    global D, RUN_KW
    if args.workload == "capm":
        spec, D = models.capm_spec(), 9
        RUN_KW = dict(use_fixed_schedule=True, n_phi=300, lam=2.1, resampling_method="systematic", n_blocks=1, n_mh_steps=3,
                      alpha=1.0, c=0.5, target=0.25, threshold_ratio=0.5)
        if args.nparts == N_PER_GPU:
            args.nparts = 200_000
    elif args.workload == "kalman":
        spec, D = models.kalman_spec(T=80, old_T=40), 13
        RUN_KW = dict(use_fixed_schedule=False, tempering_target=0.95, n_phi=100, lam=2.1, resampling_method="systematic", n_blocks=1,
                      n_mh_steps=1, alpha=0.9, c=0.5, target=0.25, threshold_ratio=0.5)
        if args.nparts == N_PER_GPU:
            args.nparts = 50_000
    else:
        spec = models.gauss_spec(D)
    seed = 1
    n_local = args.nparts
    n_total = n_local * world
    max_stages = 1500

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1 and not force_sharded:
        from smc_jl_amd import Engine

        eng = Engine(n_total, D, seed=seed, device=local_rank, max_stages=max_stages, store_history=not args.no_history)
        eng.set_model(spec)
        eng.init_from_prior()
        P0 = eng.download_cloud()            # pristine initial cloud (prior draws + log-likelihoods): the CPU baseline starts from it

        def reset():
            # every step is a whole job: the device draws the initial cloud again (same seed, same Philox streams => the same cloud,
            # bit for bit) and runs the tempering loop on it - nothing comes from the host
            eng.init_from_prior()

        def one_step(profile=False):
            reset()
            return eng.run(use_graph=(2 if profile else (1 if args.mode == "graph" else 0)), solver_passes=args.solver_passes,
                           sync_every=args.sync_every, phi_rtol=args.phi_rtol, **RUN_KW)
    else:
        # one process per GPU: equal contiguous shards, RCCL communicator bootstrapped through torch.distributed
        from smc_jl_amd import Engine, comm_unique_id

        eng = Engine(n_total, D, seed=seed, device=local_rank, max_stages=max_stages, store_history=not args.no_history,
                     n_local=n_local, gid0=rank * n_local)
        eng.set_model(spec)
        eng.init_from_prior()
        uid = [comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(rank, world, uid[0])

        def one_step(profile=False):
            eng.init_from_prior()            # every step is a whole job: each rank draws its shard again (global particle ids)
            return eng.run_sharded(solver_passes=args.solver_passes, use_graph=2 if profile else 0, **RUN_KW)

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    stages = 0
    last = None
    for _ in range(args.steps):
        last = one_step()
        stages += last["n_stages"] - 1
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = n_total * stages / dt

    out = {
        "metric": "particle-stages/sec", "value": value, "unit": "particle-stages/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("gauss%d_isotropic_adaptive_phi_n%dk_per_gpu" % (D, n_local // 1000)) if args.workload == "gauss10"
                   else ("capm_literal_fixed_schedule_3mh_n%dk_per_gpu" % (n_local // 1000) if args.workload == "capm"
                         else "lgss_kalman13_old40_new80_adaptive_phi_n%dk_per_gpu" % (n_local // 1000)),
                   "n_parts_total": n_total, "n_para": D, "tempering_target": RUN_KW.get("tempering_target", 0.97),
                   "n_phi": RUN_KW.get("n_phi", 300), "lambda": 2.1,
                   "resampling": "systematic", "n_blocks": 1, "n_mh_steps": RUN_KW["n_mh_steps"], "launch_mode": args.mode,
                   "history": not args.no_history, "parallelism": "particles sharded x%d" % world},
        "n_stages": last["n_stages"], "resamples": last["resamples"], "logmdd_gpu": last["logmdd"],
        "logmdd_exact": models.gauss_logmdd(D) if args.workload == "gauss10" else None, "solver_passes_per_stage": last.get("solver_passes", 0) / max(last["n_stages"] - 1, 1),
    }

    sharded = world > 1 or force_sharded
    prof = None
    if sharded:
        # every rank takes part in the profiled run (it holds collectives); rank 0 reports its own GPU's kernel
        prof = one_step(profile=True)
        barrier()
    if rank == 0:
        # ---- roofline of the dominant kernel (mutation): HIP events around every k_mutate launch of one more
        # identical run on the engine's stream (use_graph = 2), algorithmic bytes / mean duration.  Multi-GPU lines quote the
        # kernel of rank 0's GPU on its shard (n_local particles): per-GPU figures, like `peak`.
        if prof is None:
            prof = one_step(profile=True)
        n_k = n_local if sharded else n_total
        nl = max(prof["n_mutate_launches"], 1)
        mean_ms = prof["kernel_ms_mutate"] / nl
        bytes_per_launch = mutate_bytes_per_particle(D) * n_k          # all MH steps of a stage are fused in the one launch
        achieved = bytes_per_launch / (mean_ms * 1e-3) / 1e9 if mean_ms > 0 else 0.0
        # HBM traffic per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 x2 read correction):
        # profiles/pmc_extract.py -> profiles/r01_pmc_traffic.json, bytes per particle of this kernel x particles
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pm = json.load(f)
            k = [v for name, v in pm["kernels"].items() if "k_mutate_reg<%d," % D in name]
            if k:
                traffic = k[0]["bytes_per_particle"] * n_k
        except OSError:
            pass
        out["roofline"] = {"bound": "hbm", "kernel": ("k_mutate_reg<%d,true>" % D) if D <= 10 else "k_mutate<0>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "bytes_per_launch": bytes_per_launch, "mean_launch_us": 1e3 * mean_ms, "launches": nl}
        # The kernel is FP64-VALU bound, not HBM bound: express it against the VALU issue rate as well.
        # One VALU instruction of a wave64 occupies a SIMD for 4 cycles; instr_per_wave is the static count of the
        # straight-line kernel body from its gfx950 ISA (profiles/isa_count.py -> r01_isa_counts.json; an upper bound on
        # the dynamic count, the MH loop is unrolled for one step x one block).
        try:
            with open(os.path.join(ROOT, "profiles", "r01_isa_counts.json")) as f:
                # below 5e5 draws per stage the random numbers are drawn ahead by k_prepare_mutation's idle CUs (csrc ensure_zbuf)
                ahead = n_local * RUN_KW.get("n_mh_steps", 1) * RUN_KW.get("n_blocks", 1) <= 500000 and not os.environ.get("SMCMI_NO_RNG_AHEAD")
                isa = json.load(f).get(("k_mutate_reg<%d,true>" % D) + (" rng_ahead" if ahead else ""))
            if isa and RUN_KW.get("n_mh_steps", 1) == 1 and mean_ms > 0:
                waves = -(-n_k // 64)
                # SIMD-32: a wave64 VALU instruction issues over 2 cycles, FP64 over 4 (half rate), 32x32-bit multiplies over 8
                cyc = 4 * isa["valu_f64"] + 2 * isa["valu_other"] + 8 * isa.get("valu_int_mul", 0)
                peak = 256 * 4 * 2.4e9                # SIMD issue cycles / s
                ach = waves * cyc / (mean_ms * 1e-3)
                out["roofline"]["valu_issue"] = {"valu_instr_per_wave": isa["valu_total"], "issue_cycles_per_wave": cyc, "waves": waves,
                                                 "achieved": ach, "peak": peak, "unit": "SIMD issue cycles/s", "frac": ach / peak,
                                                 "waves_per_simd": waves / 1024.0, "rng_drawn_ahead": bool(ahead)}
        except OSError:
            pass
        if args.workload == "kalman" and mean_ms > 0:
            # config 5 is compute bound: ~3300 FP64 flops per filter step (FMA = 2; csrc/model.hpp kalman_lgss), steps = new + old
            # periods per proposal; FP64 peak of MI355X = 256 CUs x 4 SIMDs x 16 FMA lanes x 2 x 2.4 GHz = 78.6 TFLOP/s
            # (vector = matrix rate for FP64 on gfx950)
            steps = spec["lik"][2].shape[1] + (spec["old_lik"][2].shape[1] if spec["old_lik"] else 0)
            flops = 3300.0 * steps * RUN_KW["n_mh_steps"] * n_k
            tf = flops / (mean_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "k_mutate<0> / kalman_lgss (FP64 vector FMA)", "achieved": tf, "peak": 78.6,
                               "unit": "TFLOP/s", "frac": tf / 78.6, "traffic": None, "flops_per_launch": flops,
                               "mean_launch_us": 1e3 * mean_ms, "launches": nl}
        # whole-stage algorithmic bytes: 24d+96 per particle-stage, +16d+104 on resample stages
        stage_bytes = n_total * ((24 * D + 96) * (last["n_stages"] - 1) + (16 * D + 104) * last["resamples"])
        out["stage_gbs"] = stage_bytes * args.steps / dt / 1e9
        if not args.no_cpu and not sharded:          # CPU baseline: rank 0 at N = 1 only
            from oracle import oracle as orc

            orc.build()
            m = models.oracle_model(spec)
            cores = os.cpu_count() or 1
            r = orc.smc_run(m, P0, seed=seed, n_threads=cores, history=False, max_stages=max_stages, **RUN_KW)
            cpu_value = n_total * (r["n_stages"] - 1) / r["seconds"]
            out["cpu_baseline"] = {"value": cpu_value, "unit": "particle-stages/s", "cores": cores, "kind": "port",
                                   "sample": "the full workload once (n_parts=%d, %d stages, same Philox seed); OpenMP over "
                                             "particles in the mutation step only, like the reference's @distributed "
                                             "mutation (src/smc_main.jl:472-476)" % (n_total, r["n_stages"] - 1),
                                   "seconds": r["seconds"], "logmdd": r["logmdd"]}
            out["logmdd_cpu"] = r["logmdd"]
            out["logmdd_abs_err"] = abs(last["logmdd"] - r["logmdd"])
            out["gpu_over_cpu"] = value / cpu_value
    if rank == 0:
        print(json.dumps(out))
    faulthandler.cancel_dump_traceback_later()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
