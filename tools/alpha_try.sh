#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f stages %d res %d mut_us %.2f us/stage %.1f logmdd %.12f' % (d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], d['roofline']['mean_launch_us'], 1e3*d['ms_per_step']/(d['n_stages']-1), d['logmdd_gpu']))"; }
for a in 1.0 0.9; do for nb in 1 2; do for eng in 0 1; do
echo "== alpha=$a n_blocks=$nb engine=$eng"; run SMCMI_ENGINE=$eng python bench.py --steps 5 --warmup 1 --no-cpu --alpha $a --n-blocks $nb 2>/dev/null
done; done; done
echo "== 1e6 alpha 0.9 / 1.0 (engine 1)"; run python bench.py --steps 3 --warmup 1 --no-cpu --no-history --nparts 1000000 --alpha 0.9 2>/dev/null;  run python bench.py --steps 3 --warmup 1 --no-cpu --no-history --nparts 1000000 --alpha 1.0 2>/dev/null
