cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "direct_and_reduced or shard_count" 2>&1 | tail -5
bash tools/shard_rank_prof.sh big1 250000 500000 2>&1
echo "=== helpers off"
export SMCMI_E2_HELPERS=0
bash tools/shard_rank_prof.sh big1_nohelp 250000 500000 2>&1
