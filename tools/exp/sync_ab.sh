#!/bin/bash
# development: batch length (stages per host sync) against run time
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for se in 0 192 320 600; do
  for rep in 1 2; do
    python bench.py --steps 10 --warmup 2 --no-cpu --sync-every $se 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('config2 sync_every $se', '%.4g'%d['value'], '%.3f'%d['ms_per_step'], d['n_stages'])"
  done
done
for se in 0 64 128 300; do
  python bench.py --steps 5 --warmup 1 --no-cpu --no-history --nparts 1000000 --sync-every $se 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('1e6 sync_every $se', '%.4g'%d['value'], '%.3f'%d['ms_per_step'], d['n_stages'])"
done
