#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu --workload capm --nparts $N 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f stages %d res %d mut_us %.2f logmdd %.12f' % (d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], d['roofline']['mean_launch_us'], d['logmdd_gpu']))"; }
N=200000
for nb in 13 14 15 16 17 20 25; do echo "== reduced nb1=$nb"; run SMCMI_ENGINE=2 SMCMI_E2_NB1=$nb; echo "== direct nb1=$nb"; run SMCMI_E2_DIRECT_MAX=512 SMCMI_E2_NB1=$nb; done
