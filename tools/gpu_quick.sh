#!/bin/bash
# quick GPU check used while iterating: parity subset, bench line, per-kernel table (rocprofv3 kernel trace).  usage: bash tools/gpu_quick.sh <tag> [bench args]
TAG=${1:-q}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu "$@" > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value %.4g  ms/step %.3f  stages %d  mut_us %.2f  logmdd %.12f" % (d["value"], d["ms_per_step"], d["n_stages"], d["roofline"]["mean_launch_us"], d["logmdd_gpu"]))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu "$@" > /dev/null 2>&1
cd $ROOT
python profiles/summarize_rocpd.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt
rm -rf $OUT/kt
cat $OUT/kernel_stats.txt
