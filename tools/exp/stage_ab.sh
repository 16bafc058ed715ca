#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
export SMCMI_ENGINE=2 HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 SMCMI_MAILBOX=2
for N in 250000 500000; do
  export SMCMI_E2_NB1=$(( (N / 8 + 1023) / 1024 ))
  for rep in 1 2; do
  for lib in libsmcmi.so libx1.so libx2.so; do
    SMCMI_LIBRARY=$ROOT/smc.jl_amd/csrc/$lib timeout 300 python bench.py --gpus 1 --steps 6 --warmup 1 --no-cpu --no-history --nparts $N 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('N=$N $lib value %.4g ms %.3f us/stage %.1f mut_us %.2f' % (d['value'], d['ms_per_step'], 1e3*d['ms_per_step']/(d['n_stages']-1), d['roofline']['mean_launch_us']))"
  done
  done
done
