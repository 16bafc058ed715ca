#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for x in 1 0 1 0; do
SMCMI_SEG_SELECT=$x python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('SEG_SELECT=$x config2 %.4g %.3f ms stages %d res %d logmdd %.12f seg %d launches, %.2f us/stage'%(d['value'],d['ms_per_step'],d['n_stages'],d['resamples'],d['logmdd_gpu'],r['launches'],r['mean_stage_us']))"
done
SMCMI_SEG_SELECT=1 python tools/exp/fixed_small.py 2>&1 | tail -3
SMCMI_SEG_SELECT=0 python tools/exp/fixed_small.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_segments.py -x -q 2>&1 | tail -5
