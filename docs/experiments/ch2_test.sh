#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline']
print('value %.4g ms %.3f stages %d res %d us/stage %.1f logmdd %.10f kernel %s launches %s' % (d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], 1e3*d['ms_per_step']/(d['n_stages']-1), d['logmdd_gpu'], r['kernel'][:22], r.get('launches')))"; }
echo "== config2"; python bench.py --steps 8 --warmup 2 --no-cpu 2>/dev/null | grep '^{' | line
for N in 200000 250000 253952; do
  echo "== gauss10 $N"; python bench.py --nparts $N --no-history --steps 4 --warmup 1 --no-cpu 2>/dev/null | grep '^{' | line
  echo "== gauss10 $N engine 1"; SMCMI_ENGINE=1 python bench.py --nparts $N --no-history --steps 4 --warmup 1 --no-cpu 2>/dev/null | grep '^{' | line
done
echo "== capm 200000"; python bench.py --workload capm --steps 3 --warmup 1 --no-cpu 2>/dev/null | grep '^{' | line
echo "== capm 200000 engine 1"; SMCMI_ENGINE=1 python bench.py --workload capm --steps 3 --warmup 1 --no-cpu 2>/dev/null | grep '^{' | line
