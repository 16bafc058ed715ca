cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g9
echo "=== new / changed tests"
timeout 2400 python -m pytest tests/test_gpu_strict.py tests/test_gpu_config4_full.py tests/test_gpu_sweep.py tests/test_gpu_configs.py tests/test_gpu_callback.py tests/test_gpu_sharded.py -q -x 2>&1 | tail -12
echo "=== icache"
bash tools/exp/pmc_icache.sh 2>&1 | tail -30
echo "=== callback phases"
bash tools/callback_phases.sh > gpurun_out/g9/r05_callback_phases.json 2>gpurun_out/g9/cbp.err; cat gpurun_out/g9/r05_callback_phases.json | cut -c1-400
