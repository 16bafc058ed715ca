#!/bin/bash
# development: the same bench line with the current library and with variants under tools/exp/ on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --steps 8 --warmup 2 --no-cpu 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g ms %.3f mut_us %.2f' % (d['value'], d['ms_per_step'], d['roofline']['mean_launch_us']))"; }
cp smc.jl_amd/csrc/libsmcmi.so /tmp/new.so
for k in 1 2; do
echo new; run; run
for v in "$@"; do cp tools/exp/libsmcmi_$v.so smc.jl_amd/csrc/libsmcmi.so; echo $v; run; run; done
cp /tmp/new.so smc.jl_amd/csrc/libsmcmi.so
done
