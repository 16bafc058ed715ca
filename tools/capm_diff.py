"""development: first stage at which two engine settings differ on config 4's workload (run each in a subprocess: env is read once)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from smc_jl_amd import Engine
from tests import models
n = int(sys.argv[1])
spec = models.capm_spec()
e = Engine(n, 9, seed=1, max_stages=400, store_history=False)
e.set_model(spec); e.init_from_prior()
r = e.run(use_fixed_schedule=True, n_phi=int(sys.argv[2]), lam=2.1, n_mh_steps=3)
rec = e.stage_records(r["n_stages"])
P = e.download_cloud()
print("RES " + json.dumps(dict(ess=[float(x) for x in rec["ess"]], acc=[float(x) for x in rec["accept_hist"]], logmdd=r["logmdd"], res=r["resamples"], csum=float(P[:, :9].sum()))))
''' % ROOT
def run(env, n, nphi):
    p = subprocess.run([sys.executable, "-c", W, str(n), str(nphi)], env=dict(os.environ, **env), capture_output=True, text=True)
    if p.returncode: print(p.stderr[-2000:]); raise SystemExit(1)
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RES ")][-1][4:])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
nphi = int(sys.argv[2]) if len(sys.argv) > 2 else 300
a = run({"SMCMI_ENGINE": "1"}, n, nphi)
b = run({"SMCMI_E2_DIRECT_MAX": "512"}, n, nphi)
print("logmdd", a["logmdd"], b["logmdd"], "res", a["res"], b["res"])
for k in range(len(a["ess"])):
    de, da = abs(a["ess"][k] - b["ess"][k]) / max(abs(a["ess"][k]), 1e-300), abs(a["acc"][k] - b["acc"][k])
    if de > 1e-9 or da > 1e-9:
        print("first difference at record", k, "ess", a["ess"][k], b["ess"][k], "accept", a["acc"][k], b["acc"][k])
        for j in range(max(0, k - 2), min(len(a["ess"]), k + 3)): print(j, a["ess"][j], b["ess"][j], a["acc"][j], b["acc"][j])
        break
else:
    print("records agree")
