#!/bin/bash
# development: kernel table of one bench configuration.  usage: bash tools/exp/kt_one.sh <tag> <bench args...>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_$tag -o kt -- python $ROOT/bench.py --no-cpu "$@" 2>/dev/null | tail -1 | cut -c1-200
python $ROOT/profiles/summarize_rocpd.py $(find $OUT/kt_$tag -name "*.db" | head -1) > $OUT/kstats_$tag.txt
rm -rf $OUT/kt_$tag
head -14 $OUT/kstats_$tag.txt | cut -c1-175
