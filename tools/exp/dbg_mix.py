import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_segments as T
for d in (3, 6):
    cfg = dict(n=20_000, d=d, seed=4, spec_args=[d], kw=dict(use_fixed_schedule=False, tempering_target=0.95, alpha=0.9))
    a = T._run(cfg)[0]; b = T._run(cfg, {"SMCMI_SEG_SELECT": "0"})[0]
    print("d", d, [k for k in T._KEYS if a.get(k) != b.get(k)], a["n_segments"], b["n_segments"], a["logmdd_f"], b["logmdd_f"])
