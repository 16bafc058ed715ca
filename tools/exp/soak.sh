#!/bin/bash
# development: soak of the round's final build -> gpurun_out/soak/soak.log
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/soak
{
echo "== tools/stress_segments.py 40"; timeout 900 python tools/stress_segments.py 40 2>&1 | tail -12
echo "== tools/sweep_parity.py 24"; timeout 900 python tools/sweep_parity.py 24 2>&1 | tail -6
echo "== tools/sweep_shards.py 12"; timeout 900 python tools/sweep_shards.py 12 2>&1 | tail -6
} > gpurun_out/soak/soak.log 2>&1
tail -30 gpurun_out/soak/soak.log
