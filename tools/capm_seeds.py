"""Config 4: spread of the log-MDD estimate over Philox seeds, device and oracle side by side (is |gpu - cpu| on ONE seed inside the
Monte-Carlo spread of the estimator itself?).  usage: python tools/capm_seeds.py N n_seeds_gpu n_seeds_cpu"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc          # noqa: E402
from smc_jl_amd import Engine              # noqa: E402
from tests import models                   # noqa: E402
n, ng, nc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
spec = models.capm_spec(); m = models.oracle_model(spec)
kw = dict(use_fixed_schedule=True, n_phi=300, lam=2.1, n_mh_steps=3)
out = dict(n=n, gpu={}, cpu={})
for seed in range(1, ng + 1):
    e = Engine(n, 9, seed=seed, max_stages=300, store_history=False)
    e.set_model(spec); e.init_from_prior()
    P0 = e.download_cloud()
    g = e.run(**kw)
    out["gpu"][seed] = g["logmdd"]
    if seed <= nc:
        r = orc.smc_run(m, P0, seed=seed, n_threads=64, history=False, **kw)
        out["cpu"][seed] = r["logmdd"]
    print("seed %d gpu %.6f cpu %s" % (seed, g["logmdd"], ("%.6f" % out["cpu"][seed]) if seed in out["cpu"] else "-"), flush=True)
    e.close()
g, c = np.array(list(out["gpu"].values())), np.array(list(out["cpu"].values()))
print("N=%d gpu mean %.4f sd %.4f (n=%d) | cpu mean %.4f sd %.4f (n=%d) | paired |gpu-cpu| %s" % (n, g.mean(), g.std(ddof=1), g.size, c.mean(), c.std(ddof=1) if c.size > 1 else 0.0, c.size,
      np.round(np.abs(g[:c.size] - c), 4)))
out["summary"] = dict(gpu_mean=float(g.mean()), gpu_sd=float(g.std(ddof=1)), cpu_mean=float(c.mean()), cpu_sd=float(c.std(ddof=1)) if c.size > 1 else None)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "capm_seeds_n%d.json" % n), "w"))
