#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/tlf; mkdir -p $OUT; cd $ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $ROOT/tools/exp/fixed_small.py > /dev/null 2>&1)
python profiles/timeline_rocpd.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/timeline_fixed.txt
rm -rf $OUT/kt
grep -n "k3_segment" -B1 -A8 $OUT/timeline_fixed.txt | sed -n 1,60p
