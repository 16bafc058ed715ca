#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in 10000000 3000000 400000; do for nbe in 1024 512; do
  echo "== N=$n SMCMI_NB_E=$nbe"
  SMCMI_NB_E=$nbe timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-history --nparts $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f stages %d' % (d['value'], d['ms_per_step'], d['n_stages']))"
done; done
