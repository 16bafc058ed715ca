#!/bin/bash
# development: device timeline of one run of config 2 (bench.py default) -> gpurun_out/<tag>/timeline_*.txt
TAG=${1:-tl}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | grep '^{' > $OUT/bench.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1)
python profiles/timeline_rocpd.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/timeline_config2.txt
rm -rf $OUT/kt
tail -25 $OUT/timeline_config2.txt
python -c "
import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
