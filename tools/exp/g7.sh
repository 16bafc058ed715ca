cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g7
export SMCMI_ENGINE=2 HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519
for N in 250000 500000; do
echo "=== census, stage 40, N=$N"
SMCMI_E2_NB1=$(( (N / 8 + 1023) / 1024 )) SMCMI_MAILBOX=2 SMCMI_PROF2=40 timeout 300 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu --no-history --nparts $N 2>&1 | grep -A1 "census\|K2 0\|K2 mid" | grep smcmi2 | tail -5
done
unset SMCMI_ENGINE SMCMI_FORCE_SHARDED RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT
bash tools/shard_rank_prof.sh big7 250000 500000 2>&1 | grep -v "rocclr\|k_noop\|selftest\|k_init_prior\|k2_scan\|k2_pass\|k2_reduce\|k2_gather"
echo "=== full GPU suite"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
