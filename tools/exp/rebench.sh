#!/bin/bash
# development: the bench lines that read a PMC file of the same round, once more after the round's PMC files are in profiles/
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-r05}; OUT=gpurun_out/final; mkdir -p $OUT
python bench.py --steps 8 --warmup 2 2>/dev/null | tail -1 > $OUT/${R}_bench.json
python bench.py --alpha 0.9 --no-cpu --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_alpha09.json
python bench.py --workload capm --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_capm.json
python bench.py --nparts 1000000 --no-history --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_1e6.json
python bench.py --nparts 10000000 --no-history --no-cpu --steps 1 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_1e7.json
python bench.py --workload kalman --nparts 25000 --no-cpu --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_kalman_n25000.json
for f in _capm _1e6 _1e7 '' _alpha09; do python -c "
import json;d=json.load(open('$OUT/${R}_bench$f.json'));r=d['roofline'];print('$f',d['value'],d['ms_per_step'],r['kernel'][:30],r['frac'],r['traffic'],r.get('valu_frac'),r['pmc_file'])"; done
