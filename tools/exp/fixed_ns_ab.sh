#!/bin/bash
# A/B of the fixed-schedule stage enqueued without selection kernels (smcmi_run, h_note throttle) against the seven-launch stage.
# usage (GPU box): bash tools/exp/fixed_ns_ab.sh  -> gpurun_out/fixed_ns/
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/fixed_ns; mkdir -p $OUT; cd $ROOT
pick() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$1', 'ms', round(d['ms_per_step'],3), 'stages', d['n_stages'], 'resamples', d['resamples'], 'logmdd', repr(d['logmdd_gpu']), 'mut_us', round(d['roofline']['mean_launch_us'],2))"; }
for m in 0 1; do
  SMCMI_FIXED_NO_SELECT=$m python bench.py --workload capm --no-cpu --steps 3 --warmup 1 2>$OUT/err_$m.log | pick "capm200k select=$m"
done
for ra in 1 3 4; do
  SMCMI_FIXED_RUN_AHEAD=$ra python bench.py --workload capm --no-cpu --steps 3 --warmup 1 2>/dev/null | pick "capm200k run_ahead=$ra"
done
for m in 0 1; do
  SMCMI_FIXED_NO_SELECT=$m python bench.py --workload capm --nparts 1000000 --no-history --no-cpu --steps 2 --warmup 1 2>/dev/null | pick "capm1e6 select=$m"
done
