#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for x in 1 0 1 0; do
echo "SMCMI_SEG_NOTE=$x"; SMCMI_SEG_NOTE=$x python tools/exp/fixed_small.py 2>&1 | tail -3
SMCMI_SEG_NOTE=$x python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config2 %.4g %.3f'%(d['value'],d['ms_per_step']))"
done
timeout 900 python -m pytest tests/test_gpu_segments.py tests/test_gpu_sweep.py tests/test_gpu_errors.py -x -q 2>&1 | tail -3
