mkdir -p gpurun_out/g2
python -m pytest tests/test_gpu_segments.py -x -q 2>&1 | tail -15 > gpurun_out/g2/seg.log
python -m pytest tests/test_gpu_fake_rccl.py -x -q -k mismatched 2>&1 | tail -30 > gpurun_out/g2/fake.log
python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 > gpurun_out/g2/bench.json
python bench.py --steps 5 --warmup 2 --no-cpu --alpha 0.9 2>/dev/null | tail -1 > gpurun_out/g2/bench_a09.json
python tools/fixed_schedule.py > gpurun_out/g2/fixed.txt 2>&1
SMCMI_SHIFT_LAG=0 python tools/fixed_schedule.py 100000 5000 > gpurun_out/g2/fixed_lag0.txt 2>&1
SMCMI_PROF2=150 python tools/fixed_schedule.py 100000 > gpurun_out/g2/prof_fixed.txt 2>&1
SMCMI_PROF2=150 python bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/g2/prof_adaptive.txt 2>&1
tail -n 5 gpurun_out/g2/seg.log gpurun_out/g2/fake.log; cat gpurun_out/g2/fixed.txt gpurun_out/g2/fixed_lag0.txt; grep smcmi3 gpurun_out/g2/prof_fixed.txt | head -20; grep smcmi3 gpurun_out/g2/prof_adaptive.txt | head -12;  python -c "
import json
for f in ('bench','bench_a09'):
    d=json.load(open('gpurun_out/g2/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['segments'], d['roofline'].get('mean_stage_us'))
"
