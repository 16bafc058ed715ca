#!/usr/bin/env python
"""bench.py - particle-stages/sec of the SMC correction/selection/mutation loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete pass of the hot path over one batch: a full smc() run (all tempering stages) of the
10-dim isotropic-Gaussian log-likelihood with adaptive ϕ (tempering_target 0.97, n_Φ = 300, λ = 2.1), systematic
resampling, 1 block, 1 MH step - on synthetic prior draws made on the device at the start of every step (`initial_draw!`;
same seed => the same cloud in every step, nothing crosses PCIe).
  --gpus 1 : BASELINE config 2, n_parts = 100 000 on one MI355X (the headline line);
  --gpus N : BASELINE config 3, n_parts = 1 000 000 IN TOTAL sharded over N GPUs (strong scaling: 1e6 / N particles per rank,
             one rank per GPU, RCCL over xGMI; small all-gathers of block sums per stage, all-to-all-v of rows on resample stages).
value = n_parts * (n_stages - 1) * K / wall: the reference's metric (stage bracket src/smc_main.jl:378,489-490) with the
device-side initial draw inside the wall time as well; file I/O excluded.
When --gpus N > 1 is given without a torch.distributed environment (WORLD_SIZE unset) the script re-launches itself under
`python -m torch.distributed.run --nproc-per-node N`; it refuses to run with a world size other than N.
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
# The CPU-baseline leg runs OpenMP regions back to back with serial stretches in between (the reference's serial bisection):
# spinning worker threads (libgomp's default wait policy) slow the serial part 5x on a 256-thread host.  Must be in the
# environment before the first OpenMP runtime is loaded (torch brings one).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

D = 10
N_PER_GPU = 100_000            # config 2 (the --gpus 1 line)
N_TOTAL_SHARDED = 1_000_000    # config 3 (--gpus N > 1): this many particles IN TOTAL, 1e6 / N per GPU
RUN_KW = dict(use_fixed_schedule=False, tempering_target=0.97, n_phi=300, lam=2.1, resampling_method="systematic",
              n_blocks=1, n_mh_steps=1, alpha=1.0, c=0.5, target=0.25, threshold_ratio=0.5)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def mutate_bytes_per_particle(d):
    # mutate reads θ(d), ℓ, π, ℓ_old and writes θ(d), ℓ, π, ℓ_old, accept = 16 d + 56 bytes (FP64)
    return 16 * d + 56


def spawn_ranks(n_gpus, argv, environ=None, run=None):
    """`python bench.py --gpus N` outside torch.distributed (no WORLD_SIZE): re-launch this script with one rank per GPU, exactly
    as the driver does for N > 1.  Returns the launcher's exit code.  `run` is injectable for the CPU-side unit test."""
    import socket
    import subprocess

    env = dict(os.environ if environ is None else environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return (run or subprocess.call)(cmd, env=env)


def probe_julia(run=None):
    """is the reference's own runtime on this box?  `julia --version` (the C port stands in for the Distributed.jl path while
    it is not: `cpu_baseline.kind` = "port").  Returns the version line, or "unavailable"."""
    import shutil
    import subprocess

    exe = shutil.which("julia")
    if not exe:
        return "unavailable"
    try:
        p = (run or subprocess.run)([exe, "--version"], capture_output=True, text=True, timeout=30)
        line = (p.stdout or "").strip().splitlines()
        return (line[0] if line else "unavailable") + " (present, but SMC.jl and its dependencies are not installed offline: not timed)"
    except Exception:
        return "unavailable"


def resolve_world(n_gpus, environ):
    """(world, must_spawn): the world size this process runs in and whether bench.py has to launch the ranks itself.
    Raises SystemExit when the environment's world size contradicts --gpus (a silent 1-GPU run must never be reported as N)."""
    if "WORLD_SIZE" not in environ:
        return (n_gpus, n_gpus > 1)
    world = int(environ["WORLD_SIZE"])
    if world != n_gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to run (the line would misreport n_gpus)" % (n_gpus, world))
    return (world, False)


def main():
    # A collective that never completes (a rank died, a transport misbehaves) must not hold the GPUs until somebody's outer limit
    # fires: dump the Python stacks and exit after SMCMI_BENCH_WATCHDOG seconds (default 15 min; the default run takes ~1 min).
    import faulthandler

    faulthandler.dump_traceback_later(int(os.environ.get("SMCMI_BENCH_WATCHDOG", "900")), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nparts", type=int, default=0, help="particles in total (default: 100 000 at --gpus 1, 1 000 000 at --gpus N > 1)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU-oracle baseline leg")
    ap.add_argument("--no-history", action="store_true")
    ap.add_argument("--no-ref", action="store_true", help="--gpus N > 1: skip the single-GPU run of the same workload on rank 0")
    ap.add_argument("--workload", default="gauss10", choices=["gauss10", "capm", "kalman"],
                    help="gauss10 = BASELINE config 2 (the bench line); capm = config 4 (examples/capm_model, 3 MH steps, fixed schedule); kalman = config 5 (13-parameter state-space model, Kalman-filter likelihood, old + new data)")
    ap.add_argument("--solver-passes", type=int, default=0)
    ap.add_argument("--sync-every", type=int, default=0)
    ap.add_argument("--phi-rtol", type=float, default=0.0, help="adaptive-phi root tolerance (0 = library default)")
    ap.add_argument("--alpha", type=float, default=None, help="mixture weight of the proposal (default: the workload's own)")
    ap.add_argument("--n-blocks", type=int, default=None, help="random parameter blocks per MH step (default: the workload's own)")
    args = ap.parse_args()

    world, must_spawn = resolve_world(args.gpus, os.environ)
    if must_spawn:
        faulthandler.cancel_dump_traceback_later()
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch

    from smc_jl_amd.host import workloads as models

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SMCMI_BENCH_COMM=host (development / tests): the ranks share ONE GPU and the sharded driver's collectives go through the library's
    # host-mediated communicator over gloo (include/smcmi.h smcmi_comm_init_host) - the whole multi-rank code path of this script,
    # pre-flight included, on a one-GPU box.  Never what a scaling number is measured with (config.comm says which it was).
    # SMCMI_BENCH_COMM=rccl_shared (tests): the ranks share one GPU as well, but the library's RCCL branch carries the collectives (smcmi_comm_init ->
    # whatever SMCMI_RCCL_PATH names: tests/fake_rccl, the shared-memory stand-in); torch.distributed itself runs over gloo.
    host_comm = os.environ.get("SMCMI_BENCH_COMM") == "host"
    shared_gpu = host_comm or os.environ.get("SMCMI_BENCH_COMM") == "rccl_shared"
    if world > 1 and torch.cuda.device_count() < world and not shared_gpu:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    if shared_gpu:
        local_rank = 0
    red_dev = "cpu" if shared_gpu else "cuda"           # device of the small tensors torch.distributed reduces
    torch.cuda.set_device(local_rank)
    dist = None
    force_sharded = os.environ.get("SMCMI_FORCE_SHARDED") == "1"
    if world > 1 or (force_sharded and "RANK" in os.environ):
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    global D, RUN_KW
    if args.workload == "capm":
        spec, D = models.capm_spec(), 9
        RUN_KW = dict(use_fixed_schedule=True, n_phi=300, lam=2.1, resampling_method="systematic", n_blocks=1, n_mh_steps=3,
                      alpha=1.0, c=0.5, target=0.25, threshold_ratio=0.5)
        default_total = 200_000
    elif args.workload == "kalman":
        spec, D = models.kalman_spec(T=80, old_T=40), 13
        spec_old = models.kalman_spec(T=40)
        RUN_KW = dict(use_fixed_schedule=False, tempering_target=0.95, n_phi=100, lam=2.1, resampling_method="systematic", n_blocks=1,
                      n_mh_steps=1, alpha=0.9, c=0.5, target=0.25, threshold_ratio=0.5)
        default_total = 50_000
    else:
        spec = models.gauss_spec(D)
        default_total = N_PER_GPU if world == 1 else N_TOTAL_SHARDED
    if args.alpha is not None:
        RUN_KW["alpha"] = args.alpha
    if args.n_blocks is not None:
        RUN_KW["n_blocks"] = args.n_blocks
    seed = 1
    n_total = args.nparts if args.nparts > 0 else default_total
    if n_total % world:
        raise SystemExit("bench.py: n_parts = %d is not divisible by --gpus %d (equal contiguous shards)" % (n_total, world))
    n_local = n_total // world
    max_stages = 1500

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1 and not force_sharded:
        from smc_jl_amd import Engine

        eng = Engine(n_total, D, seed=seed, device=local_rank, max_stages=max_stages, store_history=not args.no_history)
        run_extra = {}
        if args.workload == "kalman":
            # Config 5 is a generalized-tempering UPDATE: the job starts from the posterior cloud of the estimation on the old vintage
            # (smc_main.jl:244-260, tempered_update_prior_weight = 0, same n_parts).  That estimation (old data only, from the prior)
            # runs once here, outside every timed region; its cloud stays resident on the device and every step restores it with a
            # device-to-device copy, re-evaluates the likelihoods on the new vintage (initialize_likelihoods!, :308) and tempers.
            eng.set_model(spec_old)
            eng.init_from_prior()
            r_old = eng.run(**RUN_KW)
            P_old = eng.download_cloud()
            ess_old = float(eng.stage_records(r_old["n_stages"])["ess"][-1])
            d_old = torch.from_numpy(np.ascontiguousarray(P_old.T)).to("cuda:%d" % local_rank)     # column-major n x R
            eng.set_model(spec)
            run_extra = dict(initial_ess=ess_old)
            P0 = P_old                           # the CPU baseline starts from the same old-vintage cloud

            def reset():
                eng.upload_cloud_from_device(d_old.data_ptr())
                eng.initialize_likelihoods()
        else:
            eng.set_model(spec)
            eng.init_from_prior()
            P0 = eng.download_cloud()            # pristine initial cloud (prior draws + log-likelihoods): the CPU baseline starts from it

            def reset():
                # every step is a whole job: the device draws the initial cloud again (same seed, same Philox streams => the same
                # cloud, bit for bit) and runs the tempering loop on it - nothing comes from the host
                eng.init_from_prior()

        def one_step(profile=False):
            reset()
            return eng.run(use_graph=(2 if profile else 0), solver_passes=args.solver_passes,
                           sync_every=args.sync_every, phi_rtol=args.phi_rtol, **run_extra, **RUN_KW)
    else:
        # one process per GPU: equal contiguous shards, RCCL communicator bootstrapped through torch.distributed
        from smc_jl_amd import Engine, comm_unique_id

        eng = Engine(n_total, D, seed=seed, device=local_rank, max_stages=max_stages, store_history=not args.no_history,
                     n_local=n_local, gid0=rank * n_local)
        run_extra = {}
        eng.set_model(spec_old if args.workload == "kalman" else spec)
        eng.init_from_prior()
        if host_comm:
            from smc_jl_amd import torch_dist_host_comm

            eng.comm_init_host(rank, world, *torch_dist_host_comm())
        else:
            uid = [comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            eng.comm_init(rank, world, uid[0])

        if args.workload == "kalman":
            # (see the single-handle branch: estimation on the old vintage once, untimed; every step is the tempered update from it)
            r_old = eng.run_sharded(**RUN_KW)
            ess_old = float(eng.stage_records(r_old["n_stages"])["ess"][-1])
            d_old = torch.from_numpy(np.ascontiguousarray(eng.download_cloud().T)).to("cuda:%d" % local_rank)
            eng.set_model(spec)
            run_extra = dict(initial_ess=ess_old)

        def one_step(profile=False):
            if args.workload == "kalman":
                eng.upload_cloud_from_device(d_old.data_ptr())
                eng.initialize_likelihoods()
            else:
                eng.init_from_prior()        # every step is a whole job: each rank draws its shard again (global particle ids)
            return eng.run_sharded(solver_passes=args.solver_passes, use_graph=2 if profile else 0, **run_extra, **RUN_KW)

    hand_over, preflight = None, None
    if world > 1 or force_sharded:
        # Pre-flight on the hardware at hand, outside every timed region: one run with the per-stage hand-overs as RCCL all-gathers and
        # one with the library's default - the peer mailbox over xGMI when every rank could map and test it (include/smcmi.h).  Both
        # paths total the same rows in the same order, so their stage counts, resample counts and log-MDD bits must be identical; if
        # they are not on any rank, or the mailbox run fails anywhere, every rank falls back to the all-gathers for the timed steps.
        user_choice = os.environ.get("SMCMI_MAILBOX")
        os.environ["SMCMI_MAILBOX"] = "0"
        ra = one_step()
        if user_choice is None:
            os.environ.pop("SMCMI_MAILBOX")
        else:
            os.environ["SMCMI_MAILBOX"] = user_choice
        ok, used = 1, 0
        try:
            rb = one_step()
            used = 1 if eng.mailbox_active() else 0
            if (rb["n_stages"], rb["resamples"], float(rb["logmdd"]).hex()) != (ra["n_stages"], ra["resamples"], float(ra["logmdd"]).hex()):
                ok = 0
        except Exception as ex:   # noqa: BLE001
            ok = 0
            sys.stderr.write("bench.py: rank %d: run with the default hand-over failed (%s): falling back to all-gathers\n" % (rank, ex))
        segs = 1 if (ok and rb.get("n_segments", 0) > 0) else 0
        flags = torch.tensor([ok, used, segs], device=red_dev, dtype=torch.int32)
        sums = flags.clone()
        if dist is not None:
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        if int(flags[0].item()) == 0:
            os.environ["SMCMI_MAILBOX"] = "0"
        hand_over = "peer mailbox (xGMI)" if int(flags[0].item()) == 1 and int(flags[1].item()) == 1 else "RCCL all-gather"
        if shared_gpu:
            hand_over = hand_over.replace("(xGMI)", "(HIP IPC, one GPU)")
            if host_comm:
                hand_over = hand_over.replace("RCCL all-gather", "host-mediated all-gather")
            hand_over += " [SMCMI_BENCH_COMM=%s: ranks share one GPU]" % os.environ["SMCMI_BENCH_COMM"]
        # what the pre-flight found, rank by rank (a fall-back on the real node must be visible in the line, not just slower):
        #   allgather_run_ok  the run with every hand-over as an all-gather completed on this many ranks (all, or the job has died above)
        #   default_run_ok    ranks on which the library's default transport completed AND reproduced that run's stage count, resample count and
        #                     log-MDD bit for bit (anything short of `ranks` => every rank falls back to all-gathers for the timed steps)
        #   mailbox_ranks     ranks whose default run handed its sums over through the peer mailbox
        #   segment_ranks     ranks whose default run ran its stages inside persistent segments that span the ranks
        preflight = {"ranks": world, "allgather_run_ok": world, "default_run_ok": int(sums[0].item()), "bits_equal": int(flags[0].item()) == 1,
                     "mailbox_ranks": int(sums[1].item()), "segment_ranks": int(sums[2].item()),
                     "mailbox_ok": int(flags[0].item()) == 1 and int(flags[1].item()) == 1,
                     "collectives": ("host functions over gloo" if host_comm else ("RCCL entry points from " + os.environ.get("SMCMI_RCCL_PATH", "librccl.so"))),
                     "timed_steps_use": hand_over}
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
        t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = n_total * stages / dt

    out = {
        "metric": "particle-stages/sec", "value": value, "unit": "particle-stages/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1),
        "higher_is_better": True, "scaling": "strong" if world > 1 else None, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (("gauss%d_isotropic_adaptive_phi_n%dk" % (D, n_total // 1000)) +
                                (" (BASELINE config 3: %d particles in total, strong scaling over %d GPUs, %d per GPU)" % (n_total, world, n_local)
                                 if world > 1 else " (BASELINE config 2)")) if args.workload == "gauss10"
                   else ("capm_literal_fixed_schedule_3mh_n%dk" % (n_total // 1000) if args.workload == "capm"
                         else "lgss_kalman13_tempered_update_old40_new80_adaptive_phi_n%s" % (("%dk" % (n_total // 1000)) if n_total % 1000 == 0 else str(n_total))),
                   "n_parts_total": n_total, "n_parts_per_gpu": n_local, "n_para": D, "tempering_target": RUN_KW.get("tempering_target", 0.97),
                   "n_phi": RUN_KW.get("n_phi", 300), "lambda": 2.1,
                   "resampling": "systematic", "n_blocks": RUN_KW["n_blocks"], "alpha": RUN_KW["alpha"], "n_mh_steps": RUN_KW["n_mh_steps"], "launch_mode": "direct",
                   "history": not args.no_history, "parallelism": "particles sharded x%d" % world,
                   "hand_over": hand_over, "preflight": preflight},
        "n_stages": last["n_stages"], "resamples": last["resamples"], "logmdd_gpu": last["logmdd"],
        # stages that ran inside persistent segments (engine 3; with several ranks: sharded segments through the peer mailbox)
        "segments": last.get("n_segments", 0), "segment_stages": last.get("segment_stages", 0),
        # blocks (= CUs: all resident for the whole launch) of a segment launch; the handle's segment state after the run (1 usable, 0 not applicable,
        # -1 residency self-test failed / a hand-over timed out: launches only); runs repeated as launches after a time-out; the stage from which a
        # fixed-schedule run fell back from lagged to exact energy shifts (0: never)
        "segment_blocks": last.get("segment_blocks", 0), "segment_state": last.get("segment_state", 0), "segment_timeouts": last.get("segment_timeouts", 0),
        "shift_fallback_stage": last.get("shift_fallback_stage", 0),
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
        nl_real = prof["n_mutate_launches"]                      # (0: every mutation of the run ran inside segments)
        nl = max(nl_real, 1)
        mean_ms = prof["kernel_ms_mutate"] / nl
        bytes_per_launch = mutate_bytes_per_particle(D) * n_k          # all MH steps of a stage are fused in the one launch
        achieved = bytes_per_launch / (mean_ms * 1e-3) / 1e9 if mean_ms > 0 else 0.0
        # Counter figures for this kernel at THIS cloud size: profiles/rNN_pmc_<workload>_n<particles per GPU>.json, written by
        # profiles/pmc_extract.py from three separate rocprofv3 --pmc passes of this very command (FETCH_SIZE; WRITE_SIZE; SQ
        # counters).  traffic = HBM bytes per launch (gfx950 x2 read correction); VALU fraction = SQ_ACTIVE_INST_VALU quad-cycles
        # x 4 / (kernel duration x 1024 SIMDs x clock).  No file for this size => null (never scaled from another size).
        import glob
        import re

        kname = ("k_mutate_reg<%d," % D) if D <= 10 else ("k2w_mutate<%d," % D)
        # (small clouds and sharded runs use engine 2's k2_mutate, a single handle with a larger cloud engine 1's k_mutate_reg;
        # n_para 11..16: k2w_mutate - engine 2's prologue in front of the generic mutation body - or, SMCMI_ENGINE=1, engine 1's k_mutate)
        kname_run = ("k2_mutate<%d,...> / k2b_mutate<%d,...> / k_mutate_reg<%d,...>" % (D, D, D)) if D <= 10 else ("k2w_mutate<%d,...>" % D)
        traffic, valu = None, None
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s_n%d.json" % (args.workload, n_k))),
                           key=lambda f: int(re.search(r"r(\d+)", os.path.basename(f)).group(1)))
        pmc_file = pmc_files[-1] if pmc_files else None
        if pmc_file:
            with open(pmc_file) as f:
                pm = json.load(f)
            k = [(name, v) for name, v in pm["kernels"].items() if kname in name or ("k2_mutate<%d," % D) in name or ("k2b_mutate<%d," % D) in name]
            if k:
                k.sort(key=lambda nv: -nv[1].get("total_bytes", 0.0))
                kname_run, traffic, valu = k[0][0].split("::")[-1], k[0][1].get("total_bytes"), k[0][1].get("valu")
        # The mutation kernel does ~30 FP64 flop per byte it moves (SURVEY ridge: ~10): it is bound by FP64 VALU issue, not by HBM.
        # `achieved`/`peak` stay the algorithmic-bytes-over-duration figure the contract defines; `bound` names the real limiter
        # and `valu` carries the counter-derived issue fraction (null without a PMC file for this size).
        out["roofline"] = {"bound": "valu" if D <= 10 else "hbm", "kernel": kname_run, "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "bytes_per_launch": bytes_per_launch, "mean_launch_us": 1e3 * mean_ms, "launches": nl_real,
                           "valu_frac": valu.get("frac") if valu else None, "valu": valu,
                           "pmc_file": os.path.relpath(pmc_file, ROOT) if pmc_file else None}
        if prof.get("n_segments", 0) > 0 and prof.get("segment_stages", 0) > 0 and prof.get("kernel_ms_segments", 0.0) > 0.0:
            # Engine 3 (csrc/stage3.hpp): runs of stages that neither resample nor need a certificate pass execute as ONE persistent
            # launch each - the dominant kernel is k3_segment and a launch is a whole run of stages, correction + moments + mutation.
            # Algorithmic bytes of a launch = (24 d + 96) bytes per particle-stage (stage without resampling) x N x the
            # stages it completed; duration = HIP events around every segment launch on the engine's stream (use_graph = 2).  The
            # cloud stays in registers inside a segment, so the HBM traffic the counters see is far BELOW the algorithmic figure
            # (history columns + one row per block and phase); the stage is bound by the latency of its two chip-wide hand-overs.
            seg_ms, seg_st, seg_n = prof["kernel_ms_segments"], prof["segment_stages"], prof["n_segments"]
            stage_b = (24 * D + 96) * n_k
            ach3 = stage_b * seg_st / (seg_ms * 1e-3) / 1e9
            traffic3, pmc3 = None, None
            if pmc_file:
                k3 = [(name, v) for name, v in pm["kernels"].items() if "k3_segment<%d," % D in name]
                if k3 and k3[0][1].get("launches"):
                    traffic3, pmc3 = k3[0][1].get("sum_total_bytes", 0.0) / k3[0][1]["launches"], k3[0][1].get("valu")      # HBM bytes of an average launch
            riding = bool(RUN_KW.get("use_fixed_schedule")) and os.environ.get("SMCMI_SHIFT_LAG", "1") != "0" and not sharded
            # `frac` stays what the contract defines (algorithmic bytes / duration / HBM peak); `bound` names the real limiter: a stage is a chain
            # of dependent work in ONE block between chip-wide hand-overs.  floor_us = that chain with hand-overs of two store->load hops and
            # nothing else (DESIGN §4b's phase table: correction row 3.8 + decision / proposal 5.7 + MH step 4.5 + mutation row 2.7 + begin 3.2
            # [adaptive; 1.3 fixed] + draws 2.4 where no wait hides them + 1.1 per hand-over): the kernel's own ceiling, not the chip's.
            floor_us = (3.8 + 2.4 + 1.3 + 5.7 + 4.5 + 1.7 + 1.1) if riding else (3.8 + 5.7 + 4.5 + 2.7 + 3.2 + 2 * 1.1)
            out["roofline"] = {"bound": "latency", "kernel": "k3_segment<%d, %s, %s> (persistent: one launch = a run of stages)" % (D, "true" if RUN_KW["alpha"] == 1.0 else "false", "true" if riding else "false"),
                               "achieved": ach3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach3 / HBM_PEAK_GBS, "traffic": traffic3,
                               "bytes_per_launch": stage_b * seg_st / seg_n, "mean_launch_us": 1e3 * seg_ms / seg_n, "launches": seg_n,
                               "stages_per_launch": seg_st / seg_n, "bytes_per_stage": stage_b, "mean_stage_us": 1e3 * seg_ms / seg_st,
                               "floor_us": floor_us, "floor_frac": floor_us / (1e3 * seg_ms / seg_st),
                               "cus_occupied": prof.get("segment_blocks", 0), "cus": 256,
                               "hand_overs_per_stage": 1 if riding else 2,
                               "note": "latency-bound at this N, not HBM-bound (`frac` is the contract's algorithmic-bytes figure; the cloud stays in "
                                       "registers, `traffic` is history columns + one row per block and phase): %s per stage (two store->load hops "
                                       "each) + the serial decision / proposal / %s work of one block; resample stages run inside the segment, "
                                       "certificate stages as engine 2's launches in front of the segment that enters at their mutation "
                                       "(%d segment launches for %d stages)" % ("ONE chip-wide hand-over" if riding else "two chip-wide hand-overs",
                                                                                "schedule" if riding else "Newton", seg_n, last["n_stages"] - 1),
                               "valu": pmc3, "pmc_file": os.path.relpath(pmc_file, ROOT) if pmc_file else None,
                               "mutation_kernel_outside_segments": {"kernel": kname_run, "mean_launch_us": 1e3 * mean_ms if nl_real else None, "launches": nl_real}}
        if args.workload == "kalman" and mean_ms > 0:
            # config 5 is compute bound: ~3300 FP64 flops per filter step (FMA = 2; csrc/model.hpp kalman_lgss), steps = new + old
            # periods per proposal; FP64 peak of MI355X = 256 CUs x 4 SIMDs x 16 FMA lanes x 2 x 2.4 GHz = 78.6 TFLOP/s
            # (vector = matrix rate for FP64 on gfx950)
            # The old vintage (40 periods) is a prefix of the data (80 periods) and the old likelihood the same model: the library
            # takes both log-likelihoods from ONE pass over the data (bit for bit what two passes give; SMCMI_NO_LIK_PREFIX=1 runs
            # two) - the flops counted are those of the filter steps actually executed.
            t_new = spec["lik"][2].shape[1]
            t_old = spec["old_lik"][2].shape[1] if spec["old_lik"] else 0
            shared = bool(spec["old_lik"]) and os.environ.get("SMCMI_NO_LIK_PREFIX", "0") in ("", "0") and \
                np.array_equal(spec["old_lik"][2], spec["lik"][2][:, :t_old])
            steps = t_new if shared else t_new + t_old
            flops = 3300.0 * steps * RUN_KW["n_mh_steps"] * n_k
            tf = flops / (mean_ms * 1e-3) / 1e12
            lanes = os.environ.get("SMCMI_KALMAN_LANES", "")
            split = lanes == "4" or (lanes != "1" and n_k <= 32768)
            wide = os.environ.get("SMCMI_ENGINE", "0") != "1"
            kk = (lambda ls: ("k2w_mutate<13, %d> (decision + proposal prologue, then the filter)" % ls) if wide else ("k_mutate<0, %d>" % ls))
            kname5 = (kk(4) + " / kalman_lgss_quad: four lanes per particle (the default up to 32 768 particles per handle)" if split else
                      kk(1) + " / kalman_lgss_wave: one thread per particle, structure values through DPP operands")
            # counter figures for the filter's kernel at this cloud size, when a PMC file exists (profiles/rNN_pmc_kalman_n<N>.json)
            traffic5, valu5 = None, None
            if pmc_file:
                k5 = [(name, v) for name, v in pm["kernels"].items() if "k_mutate<0," in name or "k2w_mutate<13," in name]
                if k5:
                    k5.sort(key=lambda nv: -nv[1].get("sum_total_bytes", nv[1].get("total_bytes", 0.0)))
                    traffic5, valu5 = k5[0][1].get("total_bytes"), k5[0][1].get("valu")
            # (the FP64 matrix instructions run on the vector unit's FP64 datapath: v_mfma_f64_16x16x4 / _4x4x4_4b reach 78 / 76 TFLOP/s alone and
            # SERIALISE with a v_fma_f64 stream of another wavefront on the same SIMD - profiles/r05_mfma_f64.json, tools/ubench/mfma_f64.hip - so
            # the filter is bound by the FP64 vector issue rate whichever instruction carries its products)
            out["roofline"] = {"bound": "valu", "kernel": kname5 + " (FP64 vector FMA; the FP64 matrix pipe shares the vector datapath on gfx950: measured, profiles/r05_mfma_f64.json)", "achieved": tf, "peak": 78.6,
                               "unit": "TFLOP/s", "frac": tf / 78.6, "traffic": traffic5, "valu_frac": valu5.get("frac") if valu5 else None, "valu": valu5,
                               "pmc_file": os.path.relpath(pmc_file, ROOT) if pmc_file else None, "flops_per_launch": flops,
                               "mean_launch_us": 1e3 * mean_ms, "launches": nl,
                               "filter_steps_per_proposal": steps, "old_data_prefix_shared": shared}
        # whole-stage algorithmic bytes: 24d+96 per particle-stage, +16d+104 on resample stages
        stage_bytes = n_total * ((24 * D + 96) * (last["n_stages"] - 1) + (16 * D + 104) * last["resamples"])
        out["stage_gbs"] = stage_bytes * args.steps / dt / 1e9
        if not args.no_cpu and not sharded:          # CPU baseline: rank 0 at N = 1 only
            # Two variants of the CPU port on all host cores (BASELINE.md §3 / SURVEY §8d-ii), each on the full workload once
            # (same cloud, same Philox seed):
            #  "faithful"  - the reference's cost structure: MvNormal re-factorised for every particle (mutation.jl:81), the
            #                adaptive-phi root by serial bisection to adjacent floats (helpers.jl:49, ~60 ESS passes per stage on
            #                ONE core), the mutation loop over all cores like `@distributed` (smc_main.jl:472-476);
            #  "optimised" - block factors hoisted out of the particle loop; the adaptive-phi root by 4-section on a persistent
            #                OpenMP team (one parallel region per solve, ~23 passes of 4 candidates instead of ~60 regions).
            # `value` is the faithful one (the ">= 10x" target is judged against it); both are reported.
            from oracle import oracle as orc

            orc.build()
            m = orc.model_from_spec(spec)
            cores = os.cpu_count() or 1
            variants = {}
            cpu_extra, P_cpu = {}, P0
            if args.workload == "kalman":            # the same update: initial cloud of the tempered update from the old-vintage cloud
                P_cpu, ess0 = orc.tempered_update_cloud(m, P0, ess_old, n_total, seed=seed)
                cpu_extra = dict(initial_ess=ess0)
            for name, var in (("faithful", 1), ("optimised", 2)):
                r = orc.smc_run(m, P_cpu, seed=seed, n_threads=cores, history=False, max_stages=max_stages, variant=var, **cpu_extra, **RUN_KW)
                variants[name] = {"value": n_total * (r["n_stages"] - 1) / r["seconds"], "seconds": r["seconds"], "cores": cores,
                                  "n_stages": r["n_stages"], "logmdd": r["logmdd"]}
            rf = variants["faithful"]
            out["cpu_baseline"] = {"value": rf["value"], "unit": "particle-stages/s", "cores": cores, "kind": "port",
                                   "sample": "the full workload once per variant (n_parts=%d, %d stages, same Philox seed, "
                                             "OMP_WAIT_POLICY=%s); value = the reference-faithful variant (per-particle "
                                             "re-factorisation mutation.jl:81, serial bisection helpers.jl:49, mutation over all cores "
                                             "like @distributed smc_main.jl:472-476)" % (n_total, rf["n_stages"] - 1,
                                                                                         os.environ.get("OMP_WAIT_POLICY", "default")),
                                   "seconds": rf["seconds"], "logmdd": rf["logmdd"], "variants": variants,
                                   "reference_julia": probe_julia()}
            out["logmdd_cpu"] = rf["logmdd"]
            out["logmdd_abs_err"] = abs(last["logmdd"] - rf["logmdd"])
            out["gpu_over_cpu"] = value / rf["value"]
            out["gpu_over_cpu_optimised"] = value / variants["optimised"]["value"]
        if world > 1 and not args.no_ref:
            # strong-scaling reference measured in the same job: the same n_total particles on rank 0's GPU alone
            from smc_jl_amd import Engine

            ref = Engine(n_total, D, seed=seed, device=local_rank, max_stages=max_stages, store_history=False)
            ref.set_model(spec)
            for it in range(3):
                if it == 1:
                    torch.cuda.synchronize()
                    tr0, st_ref = time.perf_counter(), 0
                ref.init_from_prior()
                rr = ref.run(solver_passes=args.solver_passes, sync_every=args.sync_every, phi_rtol=args.phi_rtol, **RUN_KW)
                if it >= 1:
                    st_ref += rr["n_stages"] - 1
            torch.cuda.synchronize()
            dtr = time.perf_counter() - tr0
            out["single_gpu_same_workload"] = {"value": n_total * st_ref / dtr, "ms_per_step": 1e3 * dtr / 2, "n_stages": rr["n_stages"],
                                               "logmdd": rr["logmdd"], "note": "rank 0 alone, history off, 2 timed steps"}
            out["speedup_vs_single_gpu"] = value / out["single_gpu_same_workload"]["value"]
            out["logmdd_abs_diff_vs_single_gpu"] = abs(last["logmdd"] - rr["logmdd"])
            ref.close() if hasattr(ref, "close") else None
    if rank == 0:
        print(json.dumps(out))
    faulthandler.cancel_dump_traceback_later()
    if dist is not None:
        dist.barrier()                # rank 0 may still be in its single-GPU reference run
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
