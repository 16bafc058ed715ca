#!/bin/bash
# development: one rank's share of a sharded run (1-rank RCCL communicator).  usage: bash tools/shard_rank_time.sh [N ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
export HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_FORCE_SHARDED=1
for N in ${@:-125000 250000 500000}; do
for env in "SMCMI_MAILBOX=0" "SMCMI_MAILBOX=2"; do
  echo "== N=$N $env"
  env $env timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu --no-history --nparts $N 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f stages %d res %d mut_us %.2f us/stage %.1f logmdd %.12f %s' % (d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], d['roofline']['mean_launch_us'], 1e3*d['ms_per_step']/(d['n_stages']-1), d['logmdd_gpu'], d['config'].get('hand_over')))"
done; done
