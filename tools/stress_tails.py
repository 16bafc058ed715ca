"""development: repeat sharded engine-2 runs (last-block row totals, optionally the in-process mailbox) and compare the bits with the
single-handle run - a race in the hand-over shows up as a differing hash.  usage: python tools/stress_tails.py <repeats> [world]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = r'''
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
from smc_jl_amd import Engine, run_group
from tests import models
world, reps, n, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 6
kw = dict(use_fixed_schedule=False, tempering_target=0.95, n_blocks=2, alpha=0.9)
out = []
for rep in range(reps):
    engs = []
    for r in range(world):
        e = Engine(n, d, seed=13, max_stages=1500, store_history=False, n_local=n // world, gid0=r * (n // world))
        e.set_model(models.gauss_spec(d)); e.init_from_prior(); engs.append(e)
    res = run_group(engs, **kw) if world > 1 else engs[0].run(**kw)
    cloud = np.concatenate([e.download_cloud() for e in engs], axis=0)
    out.append(hashlib.sha256(np.ascontiguousarray(cloud).tobytes()).hexdigest()[:16] + float(res["logmdd"]).hex())
    for e in engs: e.close()
print("RESULT " + json.dumps(out))
''' % ROOT
def run(world, reps, n, env):
    p = subprocess.run([sys.executable, "-c", W, str(world), str(reps), str(n)], env=dict(os.environ, **env), capture_output=True, text=True, cwd=ROOT)
    if p.returncode: print(p.stderr[-1500:]); raise SystemExit(1)
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for n in [int(x) for x in os.environ.get('STRESS_N', '40000,160000').split(',')]:
    ref = run(1, 1, n, {"SMCMI_ENGINE": "2"})[0]
    for env in ({}, {"SMCMI_MAILBOX": "1"}):
        if env and n > 40000: continue                       # (the in-process mailbox needs every handle's kernels resident at once)
        got = run(world, reps, n, env)
        bad = sum(1 for g in got if g != ref)
        print("n", n, "world", world, env, "runs", len(got), "mismatches", bad, flush=True)
