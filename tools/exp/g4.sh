cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g4
export SMCMI_ENGINE=2 HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519
for N in 250000; do
echo "=== phase stamps + census, stage 40, N=$N"
SMCMI_E2_NB1=$(( (N / 8 + 1023) / 1024 )) SMCMI_MAILBOX=2 SMCMI_PROF2=40 timeout 300 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu --no-history --nparts $N 2>&1 | grep "smcmi2" | tail -3
done
unset SMCMI_ENGINE SMCMI_FORCE_SHARDED RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT
echo "=== callback tests"
timeout 900 python -m pytest tests/test_gpu_callback.py -x -q 2>&1 | tail -5
echo "=== c_abi_callback: chunks default / 1"
gcc -O2 -std=c99 -ffp-contract=off -I include -o examples/c_abi_callback examples/c_abi_callback.c -L smc.jl_amd/csrc -lsmcmi -lm -Wl,-rpath,$PWD/smc.jl_amd/csrc
for rep in 1 2; do LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback 2>&1 | tee gpurun_out/g4/callback_c.json | tail -2; done
SMCMI_CB_CHUNKS=1 LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback 2>&1 | tee gpurun_out/g4/callback_c_serial.json | tail -2
SMCMI_CB_CHUNKS=4 LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback 2>&1 | tail -2
SMCMI_CB_CHUNKS=16 LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback 2>&1 | tail -2
echo "=== kalman tests"
timeout 1200 python -m pytest tests/test_gpu_kalman.py -x -q 2>&1 | tail -15
