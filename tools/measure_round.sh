#!/bin/bash
# End-of-round measurement pipeline (one MI355X): every file profiles/README.md lists, into gpurun_out/final/.
# usage (GPU box): bash tools/measure_round.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
python bench.py --steps 8 --warmup 2 2>/dev/null | tail -1 > $OUT/bench.json
python bench.py --nparts 1000000 --no-history --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_1e6.json
python bench.py --nparts 10000000 --no-history --no-cpu --steps 1 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_1e7.json
python bench.py --nparts 100000000 --no-history --no-cpu --steps 1 --warmup 0 2>/dev/null | tail -1 > $OUT/bench_1e8.json
python bench.py --workload capm --no-cpu --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_capm.json
python bench.py --workload kalman --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_kalman.json
python tools/config2_seeds.py > $OUT/config2_seeds.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | tail -1 > $OUT/bench_under_rocprof.json
rocprofv3 --kernel-trace --stats -d $OUT/kt7 -o kt7 -- python $ROOT/bench.py --nparts 10000000 --no-history --no-cpu --steps 1 --warmup 0 2>/dev/null | tail -1 > $OUT/bench_1e7_under_rocprof.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -o f -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -o w -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
cd $ROOT
python profiles/summarize_rocpd.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.txt
python profiles/summarize_rocpd.py $(find $OUT/kt7 -name "*.db" | head -1) > $OUT/kernel_stats_n1e7.txt
python profiles/pmc_extract.py $(find $OUT/pmc_f -name "*.db" | head -1) $(find $OUT/pmc_w -name "*.db" | head -1) 100000 > $OUT/pmc_traffic.json
rm -rf $OUT/kt $OUT/kt7 $OUT/pmc_f $OUT/pmc_w
ls -la $OUT
