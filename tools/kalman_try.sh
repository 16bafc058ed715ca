#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for env in "X=1" "SMCMI_NO_LIK_PREFIX=1"; do
echo "== $env"
env $env python bench.py --steps 3 --warmup 1 --no-cpu --workload kalman 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f stages %d res %d mut_us %.2f TF %.2f frac %.3f logmdd %.12f steps %s' % (d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], d['roofline']['mean_launch_us'], d['roofline']['achieved'], d['roofline']['frac'], d['logmdd_gpu'], d['roofline'].get('filter_steps_per_proposal')))"
done
timeout 900 python -m pytest tests/test_gpu_kalman.py tests/test_gpu_configs.py -q -x -m gpu -k "kalman or config5" 2>&1 | tail -3
