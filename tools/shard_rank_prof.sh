#!/bin/bash
# development: one rank's share of a sharded run with its kernel table (1-rank RCCL communicator, peer mailbox forced: the system-scope
# hand-overs a multi-GPU run uses).  usage (GPU box): bash tools/shard_rank_prof.sh <tag> [N ...]   -> gpurun_out/<tag>/
TAG=${1:-shard}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
# SMCMI_ENGINE=2: a 1-rank communicator takes the path several ranks take (run2_impl, reduced geometry beyond 131 072 particles per rank)
export SMCMI_ENGINE=${SMCMI_ENGINE:-2}
export HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f stages %d res %d mut_us %.2f us/stage %.1f logmdd %.12f %s' % (d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], d['roofline']['mean_launch_us'], 1e3*d['ms_per_step']/(d['n_stages']-1), d['logmdd_gpu'], d['config'].get('hand_over')))"; }
for N in ${@:-250000 500000}; do
  # a rank of a G-GPU run of 10^6 particles holds 8 / G virtual shards of 125 000 particles - 128 correction rows of 1 024 particles each; the
  # same particles as ONE rank's whole population are 8 virtual shards of N / 8: cut them into as many rows (SMCMI_E2_NB1) so that K1 has the
  # rank's block count and particles per block
  export SMCMI_E2_NB1=$(( (N / 8 + 1023) / 1024 ))
  for mb in 2 0; do
    echo "== N=$N SMCMI_MAILBOX=$mb SMCMI_E2_NB1=$SMCMI_E2_NB1 ${EXTRA_ENV}"
    SMCMI_MAILBOX=$mb timeout 300 python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu --no-history --nparts $N 2>/dev/null | grep '^{' | tee $OUT/bench_n${N}_mb$mb.json | line
  done
  (cd /tmp && export TMPDIR=/tmp && SMCMI_MAILBOX=2 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$N -o kt -- python $ROOT/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu --no-history --nparts $N > /dev/null 2>&1)
  python profiles/summarize_rocpd.py $(find $OUT/kt_$N -name "*.db" | head -1) > $OUT/kernel_stats_n$N.txt
  python profiles/gaps_rocpd.py $(find $OUT/kt_$N -name "*.db" | head -1) 200 > $OUT/kernel_gaps_n$N.txt
  rm -rf $OUT/kt_$N
  head -14 $OUT/kernel_stats_n$N.txt | cut -c1-60,75-140
  cat $OUT/kernel_gaps_n$N.txt
done
