"""One rank of a multi-PROCESS sharded run on ONE GPU (tests/test_gpu_multiproc.py, tests/test_gpu_fake_rccl.py): the product driver -
Engine.run_sharded -> smcmi_run_sharded -> csrc/run2.hpp run2_impl - with the host-mediated communicator over torch.distributed / gloo
or with the RCCL entry points (cfg["comm"] = "rccl"), the peer mailbox mapped between the processes through real hipIpcOpenMemHandle.  Writes its shard's results to <out>/rank<r>.json + .npy.

usage: python -m tests.mp_shard_worker <rank> <world> <port> <out_dir> <json config>"""
import hashlib
import json
import os
import sys

import numpy as np


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    cfg = json.loads(sys.argv[5])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smc_jl_amd.host import distributed as shd
        from tests import models

        spec = getattr(models, cfg.get("spec", "gauss_spec"))(*cfg.get("spec_args", []))
        n, d, seed = cfg["n"], cfg["d"], cfg["seed"]
        closures = {}
        if cfg.get("closure"):                         # the likelihood as a HOST closure on every shard (smc(loglikelihood::Function, ...), parallel = true)
            closures = models.gauss_closures(spec, tempered=cfg["closure"] == "tempered")
            spec = dict(spec, lik=("host_callback", [], None, None), old_lik=("host_callback", [], None, None) if cfg["closure"] == "tempered" else None)
        eng = shd.open_shard(spec, n, d, rank, world, seed=seed, device=0, max_stages=cfg.get("max_stages", 1500), store_history=False)
        for which, fn in enumerate(closures.get("fns", [])):
            eng.set_likelihood_callback(fn, which=which)
        eng.init_from_prior()
        if cfg.get("closure") == "tempered":
            eng.eval_cloud_callback(which=1, column=d + 2)      # old_loglh of the initial cloud (a bridge from prior draws)
        # "host": the host-mediated communicator over gloo; "rccl": smcmi_comm_init - whatever library SMCMI_RCCL_PATH names (the tests: the
        # shared-memory stand-in tests/fake_rccl, so that the driver's RCCL branch runs with ranks that share the one GPU)
        shd.connect(eng, rank, world, comm=cfg.get("comm", "host"))
        runs = []
        for rep in range(cfg.get("reps", 1)):
            if rep:
                eng.init_from_prior()
            kw = dict(cfg["kw"])
            stop = kw.pop("pause_at", 0)
            if stop:                                   # pause at a save point, then continue in place (smc_main.jl:499-507)
                r = shd.run(eng, stop_after_stage=stop, **kw)
                assert r["paused"], r
                r = shd.run(eng, continue_run=True, **kw)
            else:
                r = shd.run(eng, **kw)
            rec = eng.stage_records(r["n_stages"])
            cloud = eng.download_cloud()
            runs.append(dict(n_stages=r["n_stages"], resamples=r["resamples"], logmdd=float(r["logmdd"]).hex(),
                             schedule=hashlib.sha256(np.ascontiguousarray(rec["schedule"]).tobytes()).hexdigest(),
                             ess=hashlib.sha256(np.ascontiguousarray(rec["ess"]).tobytes()).hexdigest(),
                             accept=hashlib.sha256(np.ascontiguousarray(rec["accept_hist"]).tobytes()).hexdigest(),
                             stalls=[r.get("solver_stalls", 0), r.get("select_stalls", 0), r.get("spec_stalls", 0)],
                             mailbox=bool(eng.mailbox_active()), seconds=r["seconds"], segments=r.get("n_segments", 0),
                             segment_stages=r.get("segment_stages", 0), shift_fallback_stage=r.get("shift_fallback_stage", 0)))
            if cfg.get("full_records"):
                runs[-1]["schedule_values"] = [float(x) for x in rec["schedule"]]
        np.save(os.path.join(out, "cloud%d.npy" % rank), cloud)
        with open(os.path.join(out, "rank%d.json" % rank), "w") as f:
            json.dump(runs, f)
        eng.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
