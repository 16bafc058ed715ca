#!/bin/bash
# development: config 4 with the first 0 / 1 / 2 / 3 of its 3 proposals per particle drawn ahead in the set-up launch.  usage: bash tools/exp/rng_part_ab.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
pick() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$1', 'ms', round(d['ms_per_step'],3), 'stages', d['n_stages'], 'logmdd', repr(d['logmdd_gpu']), 'mut_us', round(d['roofline']['mean_launch_us'],2))"; }
for rep in 1 2; do
for part in 0 250000 500000 700000; do
  SMCMI_RNG_AHEAD_PART=$part python bench.py --workload capm --no-cpu --steps 3 --warmup 1 2>/dev/null | pick "capm200k part=$part"
done
done
SMCMI_RNG_AHEAD_PART=0 python bench.py --no-cpu --steps 3 --warmup 1 2>/dev/null | pick "config2"
SMCMI_RNG_AHEAD_PART=0 python bench.py --nparts 1000000 --no-history --no-cpu --steps 2 --warmup 1 2>/dev/null | pick "gauss 1e6"
