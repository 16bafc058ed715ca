#!/bin/bash
# development: engine 1's correction grid (rows the prepare block totals) at N = 1e6
cd ${GRAFT_REPO_ROOT:-/root/repo}
for nbe in 1024 768 512 256; do
  echo "== SMCMI_NB_E=$nbe"
  SMCMI_NB_E=$nbe timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --nparts 1000000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f stages %d logmdd %.12f' % (d['value'], d['ms_per_step'], d['n_stages'], d['logmdd_gpu']))"
done
