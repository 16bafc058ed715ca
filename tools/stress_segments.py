"""development: repeat single-handle runs through engine 3's segments (tagged-granule hand-overs between workers and gatherers) and
compare the bits with a run of launches only (SMCMI_ENGINE3=0) - a race in a hand-over shows up as a differing hash or a time-out.
usage: python tools/stress_segments.py <repeats> [n ...]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = r'''
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
from smc_jl_amd import Engine
from tests import models
reps, n, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kw = json.loads(sys.argv[4])
e = Engine(n, d, seed=13, max_stages=1500, store_history=False)
e.set_model(models.gauss_spec(d))
out = []
for rep in range(reps):
    e.init_from_prior()
    res = e.run(**kw)
    out.append(hashlib.sha256(np.ascontiguousarray(e.download_cloud()).tobytes()).hexdigest()[:16] + float(res["logmdd"]).hex() + ":%%d" %% res["n_segments"])
print("RESULT " + json.dumps(out))
''' % ROOT
def run(reps, n, d, kw, env):
    p = subprocess.run([sys.executable, "-c", W, str(reps), str(n), str(d), json.dumps(kw)], env=dict(os.environ, **env), capture_output=True, text=True, cwd=ROOT)
    if p.returncode: print(p.stderr[-1500:]); raise SystemExit(1)
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sizes = [int(x) for x in sys.argv[2:]] or [100000, 122880, 20000, 4099]
for n in sizes:
    for d, kw in ((10, dict(use_fixed_schedule=False, tempering_target=0.97)), (6, dict(use_fixed_schedule=False, tempering_target=0.95, n_blocks=2, alpha=0.9, n_mh_steps=2)),
                  (3, dict(use_fixed_schedule=True, n_phi=100))):
        ref = run(1, n, d, kw, {"SMCMI_ENGINE3": "0"})[0].split(":")[0]
        got = run(reps, n, d, kw, {})
        bad = sum(1 for g in got if g.split(":")[0] != ref)
        print("n", n, "d", d, "runs", len(got), "segments per run", got[0].split(":")[1], "mismatches", bad, flush=True)
