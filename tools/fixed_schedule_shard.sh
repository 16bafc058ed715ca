#!/bin/bash
# One rank's share of a fixed-schedule run of config 3's cloud on 8 GPUs (125 000 particles, the reference's default schedule: 300 stages),
# sharded segments with the peer mailbox forced on: ms per run with ONE hand-over per stage (riding) and with exact shifts (two).
# usage (GPU box): bash tools/fixed_schedule_shard.sh [n]
N=${1:-125000}
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for lag in 1 0; do
    d=$(mktemp -d)
    SMCMI_SHIFT_LAG=$lag SMCMI_MAILBOX=2 python -m tests.mp_shard_worker 0 1 2955$lag $d '{"n":'$N',"d":10,"seed":1,"kw":{"use_fixed_schedule":true,"n_phi":300},"reps":5}' 2>/dev/null
    python -c "
import json; r=json.load(open('$d/rank0.json')); b=min(x['seconds'] for x in r[1:])
print(json.dumps(dict(workload='one rank of a sharded fixed-schedule run (n_phi=300), mailbox forced on', n=$N, shift_lag=$lag, ms_per_run=1e3*b, us_per_stage=1e6*b/(r[-1]['n_stages']-1),
                      n_stages=r[-1]['n_stages'], resamples=r[-1]['resamples'], segments=r[-1]['segments'], segment_stages=r[-1]['segment_stages'], mailbox=r[-1]['mailbox'])))"
    rm -rf $d
done
