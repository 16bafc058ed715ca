"""development: which device allocation does a test read before writing?  Binary search over SMCMI_POISON_LO/HI.
usage: python tools/poison_bisect.py <pytest node id> [upper bound]"""
import os, subprocess, sys
node, hi = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400
def fails(lo, hi):
    env = dict(os.environ, SMCMI_POISON_ALLOC="1", SMCMI_POISON_LO=str(lo), SMCMI_POISON_HI=str(hi))
    p = subprocess.run([sys.executable, "-m", "pytest", node, "-q", "-x", "-m", "gpu", "--timeout", "300"], env=env, capture_output=True, text=True)
    return p.returncode != 0
lo = 0
if not fails(lo, hi): print("does not fail with allocations [0, %d) poisoned" % hi); raise SystemExit
while hi - lo > 1:
    mid = (lo + hi) // 2
    if fails(lo, mid): hi = mid
    else: lo = mid
    print("range", lo, hi, flush=True)
print("culprit allocation #", lo)
env = dict(os.environ, SMCMI_POISON_ALLOC="2", SMCMI_POISON_LO=str(lo), SMCMI_POISON_HI=str(hi))
p = subprocess.run([sys.executable, "-m", "pytest", node, "-q", "-x", "-m", "gpu", "--timeout", "300"], env=env, capture_output=True, text=True)
print([l for l in (p.stdout + p.stderr).splitlines() if "poisoned allocation" in l])
