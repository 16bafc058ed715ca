#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for rep in 1 2; do
python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('config2', '%.4g'%d['value'], '%.3f'%d['ms_per_step'], d['n_stages'], d['resamples'], '%.12f'%d['logmdd_gpu'], d['roofline']['mean_launch_us'])"
done
SMCMI_PROF2=150 python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | grep smcmi3 | tail -21
timeout 900 python -m pytest tests/test_gpu_segments.py -x -q 2>&1 | tail -3
