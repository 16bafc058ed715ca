#!/bin/bash
# Phase profile of the host-closure path (config 2 through examples/c_abi_callback.c): where a stage's wall time goes on the calling thread,
# for the batch crossing PCIe at once (SMCMI_CB_CHUNKS=1: the serial order of rounds 1-4) and in chunks (the default), with the example's
# callback on 1 and on 8 threads.  usage (GPU box): bash tools/callback_phases.sh > profiles/rNN_callback_phases.json
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
build() { gcc -O2 -std=c99 -ffp-contract=off -fopenmp -DCB_THREADS=$1 -I include -o examples/c_abi_callback examples/c_abi_callback.c -L smc.jl_amd/csrc -lsmcmi -lm -Wl,-rpath,$ROOT/smc.jl_amd/csrc; }
run() { env "$@" OMP_WAIT_POLICY=ACTIVE OMP_PROC_BIND=close LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback 2>/dev/null | head -1; }
best() { # best of three runs by particle_stages_per_s
  for r in 1 2 3; do run "$@"; done | python -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
rows.sort(key=lambda d:-d['callback']['particle_stages_per_s'])
print(json.dumps(rows[0]['callback']))"; }
echo "{"
echo " \"what\": \"examples/c_abi_callback.c, config 2 (N = 100 000, d = 10, 256 stages): ms per stage on the calling thread by phase (smcmi_callback_phases); best of 3 runs each\","
build 1
echo " \"whole_batch_at_once_callback_1_thread\": $(best SMCMI_CB_CHUNKS=1),"
echo " \"chunked_callback_1_thread\": $(best X=1),"
build 8
echo " \"whole_batch_at_once_callback_8_threads\": $(best SMCMI_CB_CHUNKS=1),"
echo " \"chunked_callback_8_threads\": $(best X=1),"
echo " \"round_4\": {\"particle_stages_per_s\": 7.594e7, \"ms_per_stage\": 1.317, \"note\": \"profiles/r04_callback_c.json: propose -> D2H -> sync -> single-threaded pack -> callback (row-wise loop) -> scatter -> H2D -> accept, strictly serial\"}"
echo "}"
