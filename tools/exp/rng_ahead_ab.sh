#!/bin/bash
# development: where drawing the mutation's random numbers ahead (in the set-up launch's idle CUs) stops paying.  usage: bash tools/exp/rng_ahead_ab.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
pick() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$1', 'ms', round(d['ms_per_step'],3), 'stages', d['n_stages'], 'logmdd', repr(d['logmdd_gpu']), 'mut_us', round(d['roofline']['mean_launch_us'],2))"; }
for mx in 500000 100000000; do
  SMCMI_RNG_AHEAD_MAX=$mx python bench.py --workload capm --no-cpu --steps 3 --warmup 1 2>/dev/null | pick "capm200k ahead_max=$mx"
  SMCMI_RNG_AHEAD_MAX=$mx python bench.py --nparts 600000 --no-history --no-cpu --steps 2 --warmup 1 2>/dev/null | pick "gauss10 6e5 ahead_max=$mx"
  SMCMI_RNG_AHEAD_MAX=$mx python bench.py --nparts 1000000 --no-history --no-cpu --steps 2 --warmup 1 2>/dev/null | pick "gauss10 1e6 ahead_max=$mx"
done
