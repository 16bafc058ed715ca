#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --workload capm --nparts $N 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f stages %d res %d mut_us %.2f logmdd %.12f' % (d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], d['roofline']['mean_launch_us'], d['logmdd_gpu']))"; }
for N in 200000; do for e in "SMCMI_ENGINE=1" "SMCMI_ENGINE=2" "SMCMI_ENGINE=2 SMCMI_E2_NO_TAIL=1"; do echo "== N=$N $e"; run $e; done; done
