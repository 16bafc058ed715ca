#!/bin/bash
# development: the bench lines of two builds of the library side by side.  usage: bash tools/ab_lib.sh <other .so> [workload nparts ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
OTHER=$1; shift
run() { env "$@" timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu --workload $WL --nparts $N 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('value %.4g ms %.3f stages %d res %d kernel_us %.2f stage_us %s logmdd %.12f' % (d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], r['mean_launch_us'], r.get('mean_stage_us'), d['logmdd_gpu']))"; }
set -- ${@:-gauss10 100000}
while [ $# -ge 2 ]; do WL=$1; N=$2; shift 2
  for rep in 1 2; do
  echo "== $WL N=$N default"; run X=1
  echo "== $WL N=$N $OTHER"; run SMCMI_LIBRARY=$PWD/$OTHER
  done
done
