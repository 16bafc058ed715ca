"""development: randomised configurations, several in-process shards (optionally through the peer mailbox) against one handle - the
bits (stage count, resamples, log-MDD, cloud hash) must agree.  usage: python tools/sweep_shards.py <seed> <trials>"""
import hashlib, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = r'''
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
from smc_jl_amd import Engine, run_group
from tests import models
world, n, d, seed, kw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), json.loads(sys.argv[5])
engs = []
for r in range(world):
    e = Engine(n, d, seed=seed, max_stages=1500, store_history=False, n_local=n // world, gid0=r * (n // world))
    e.set_model(models.gauss_spec(d, sigma=0.4)); e.init_from_prior(); engs.append(e)
try:
    res = run_group(engs, **kw) if world > 1 else engs[0].run(**kw)
    cloud = np.concatenate([e.download_cloud() for e in engs], axis=0)
    out = dict(n_stages=res["n_stages"], resamples=res["resamples"], logmdd=float(res["logmdd"]).hex(), cloud=hashlib.sha256(np.ascontiguousarray(cloud).tobytes()).hexdigest()[:16])
except Exception as ex:
    out = dict(error=str(ex)[:40])
print("RESULT " + json.dumps(out))
''' % ROOT
def run(world, n, d, seed, kw, env):
    p = subprocess.run([sys.executable, "-c", W, str(world), str(n), str(d), str(seed), json.dumps(kw)], env=dict(os.environ, **env), capture_output=True, text=True, cwd=ROOT)
    if p.returncode: return dict(crash=p.stderr[-300:])
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
SIZES = [int(x) for x in os.environ.get("SWEEP_SIZES", "16384,32768,65536").split(",")]
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    d = int(rs.randint(2, 11))
    nb = int(rs.randint(1, min(d, 3) + 1))
    while ((d + nb - 1) // nb) * (nb - 1) >= d: nb -= 1
    kw = dict(n_blocks=nb, n_mh_steps=int(rs.randint(1, 3)), alpha=float(rs.choice([1.0, 0.9, 0.5])), use_fixed_schedule=bool(rs.randint(0, 2)),
              n_phi=int(rs.choice([30, 60])), tempering_target=float(rs.choice([0.9, 0.95])), resampling_method=str(rs.choice(["systematic", "multinomial"])),
              threshold_ratio=float(rs.choice([0.5, 0.8])))
    n, seed, world = int(rs.choice(SIZES)), int(rs.randint(1, 1000)), int(rs.choice([2, 4, 8]))
    while n % world: world //= 2                       # (SWEEP_SIZES: 2 x odd and 4 x odd sizes reach the cuts into two and four virtual shards)
    ref = run(1, n, d, seed, kw, {"SMCMI_ENGINE": "2"})
    a = run(world, n, d, seed, kw, {})
    b = run(world, n, d, seed, kw, {"SMCMI_MAILBOX": "1"})
    ok = ref == a == b
    bad += 0 if ok else 1
    print(json.dumps(dict(trial=trial, ok=ok, world=world, n=n, d=d, **kw, ref=ref if not ok else ref.get("n_stages"), a=a if not ok else None, b=b if not ok else None)), flush=True)
print("bad", bad)
