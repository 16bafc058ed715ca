cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g5
export SMCMI_ENGINE=2 HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519
for N in 250000 500000; do
echo "=== census, stage 40, N=$N"
SMCMI_E2_NB1=$(( (N / 8 + 1023) / 1024 )) SMCMI_MAILBOX=2 SMCMI_PROF2=40 timeout 300 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu --no-history --nparts $N 2>&1 | grep "census" | tail -2
done
unset SMCMI_ENGINE SMCMI_FORCE_SHARDED RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT
echo "=== callback tests"
timeout 900 python -m pytest tests/test_gpu_callback.py tests/test_gpu_multiproc.py -x -q 2>&1 | tail -5
echo "=== c_abi_callback"
for th in 1 4 8; do
gcc -O2 -std=c99 -ffp-contract=off -fopenmp -DCB_THREADS=$th -I include -o examples/c_abi_callback examples/c_abi_callback.c -L smc.jl_amd/csrc -lsmcmi -lm -Wl,-rpath,$PWD/smc.jl_amd/csrc
for rep in 1 2; do OMP_WAIT_POLICY=ACTIVE LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback 2>&1 | tee gpurun_out/g5/callback_c_t$th.json | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['callback']; print($th, d['particle_stages_per_s'], d['calls'], d['ms_per_stage'], {k:round(v,4) for k,v in d['phases_ms_per_stage'].items()})"; done
done
SMCMI_CB_CHUNKS=4 OMP_WAIT_POLICY=ACTIVE LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback 2>&1 | head -1 | cut -c150-700
SMCMI_CB_CHUNKS=1 OMP_WAIT_POLICY=ACTIVE LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback 2>&1 | tee gpurun_out/g5/callback_c_serial.json | head -1 | cut -c150-700
