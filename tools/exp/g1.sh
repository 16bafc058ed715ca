mkdir -p gpurun_out/g1
python -m pytest tests/test_gpu_segments.py -x -q 2>&1 | tail -15 > gpurun_out/g1/seg.log
python -m pytest tests/test_gpu_fake_rccl.py -x -q 2>&1 | tail -30 > gpurun_out/g1/fake.log
python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 > gpurun_out/g1/bench.json
python bench.py --steps 5 --warmup 2 --no-cpu --alpha 0.9 2>/dev/null | tail -1 > gpurun_out/g1/bench_a09.json
python tools/fixed_schedule.py > gpurun_out/g1/fixed.txt 2>&1
SMCMI_SHIFT_LAG=0 python tools/fixed_schedule.py 100000 5000 > gpurun_out/g1/fixed_lag0.txt 2>&1
tail -5 gpurun_out/g1/seg.log gpurun_out/g1/fake.log; cat gpurun_out/g1/fixed.txt gpurun_out/g1/fixed_lag0.txt; python -c "
import json
for f in ('bench','bench_a09'):
    d=json.load(open('gpurun_out/g1/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['segments'], d['roofline'].get('mean_stage_us'))
"
