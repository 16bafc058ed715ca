"""development: clouds of two engine settings after a given stage of config 4's workload"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from smc_jl_amd import Engine
from tests import models
n, stop, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
e = Engine(n, 9, seed=1, max_stages=400, store_history=False)
e.set_model(models.capm_spec()); e.init_from_prior()
r = e.run(use_fixed_schedule=True, n_phi=300, lam=2.1, n_mh_steps=3, stop_after_stage=stop)
np.save(out, e.download_cloud())
''' % ROOT
n, stop = int(sys.argv[1]), int(sys.argv[2])
outs = []
for env in ({"SMCMI_ENGINE": "1"}, {"SMCMI_E2_DIRECT_MAX": "512"}, {"SMCMI_ENGINE": "2"}):
    f = tempfile.mktemp(suffix=".npy")
    p = subprocess.run([sys.executable, "-c", W, str(n), str(stop), f], env=dict(os.environ, **env), capture_output=True, text=True)
    if p.returncode: print(p.stderr[-2000:]); raise SystemExit(1)
    outs.append(np.load(f))
for k in (1, 2):
    d = np.any(np.abs(outs[0] - outs[k]) > 1e-9 * (1 + np.abs(outs[0])), axis=1)
    rows = np.nonzero(d)[0]
    print("setting", k, "rows differing from engine 1:", rows.size, rows[:10])
    if rows.size:
        i = rows[0]
        print(outs[0][i]); print(outs[k][i])
        if i > 0: print("row above:", outs[0][i - 1][:3], outs[k][i - 1][:3])
