cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for n in 12500 25000 50000; do
echo "== n=$n"
python $ROOT/bench.py --workload kalman --nparts $n --no-cpu --steps 3 --warmup 1 2>$ROOT/gpurun_out/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['n_stages'], d['resamples'], d['logmdd_gpu'], r['frac'], r['mean_launch_us'])" || tail -5 $ROOT/gpurun_out/err.log
done
cd $ROOT && timeout 900 python -m pytest tests/test_gpu_kalman.py tests/test_gpu_configs.py -x -q -k "kalman or config5 or fifty or lanes" 2>&1 | tail -3
