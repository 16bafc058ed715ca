cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
(cd tools/ubench && ./mfma_f64 > ../../gpurun_out/g2/mfma_f64.json 2>&1; cat ../../gpurun_out/g2/mfma_f64.json)
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "direct_and_reduced or shard_count" 2>&1 | tail -3
bash tools/shard_rank_prof.sh big2 250000 500000 2>&1
echo "=== phase stamps, stage 40, N=250000"
export SMCMI_ENGINE=2 HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519
SMCMI_E2_NB1=31 SMCMI_MAILBOX=2 SMCMI_PROF2=40 timeout 300 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu --no-history --nparts 250000 2>&1 | grep "smcmi2" | tail -8
echo "=== phase stamps, stage 40, N=500000"
SMCMI_E2_NB1=62 SMCMI_MAILBOX=2 SMCMI_PROF2=40 timeout 300 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu --no-history --nparts 500000 2>&1 | grep "smcmi2" | tail -8
