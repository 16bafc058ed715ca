#!/bin/bash
# development: single-handle clouds above 131 072 particles on engine 1 (default) against engine 2's large-shard stage (SMCMI_ENGINE=2)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f stages %d res %d us/stage %.1f logmdd %.10f' % (d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], 1e3*d['ms_per_step']/(d['n_stages']-1), d['logmdd_gpu']))"; }
for rep in 1 2; do
for e in 0 2; do
  echo "== capm 200000 SMCMI_ENGINE=$e"; SMCMI_ENGINE=$e python bench.py --workload capm --steps 3 --warmup 1 --no-cpu 2>/dev/null | grep '^{' | line
done; done
for N in 200000 250000 400000 500000 1000000; do
for e in 0 2; do
  echo "== gauss10 $N SMCMI_ENGINE=$e"; SMCMI_ENGINE=$e python bench.py --nparts $N --no-history --steps 4 --warmup 1 --no-cpu 2>/dev/null | grep '^{' | line
done; done
python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | grep '^{' | line
