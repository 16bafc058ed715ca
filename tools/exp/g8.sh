cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g8
echo "=== RNG-ahead A/B"
for ra in 0 1; do
export SMCMI_NO_RNG_AHEAD=$ra
bash tools/shard_rank_prof.sh big8_nora$ra 250000 500000 2>&1 | grep "^value\|^==\|k2b_mutate\|k2_correct" | grep -v "MAILBOX=0" 
done
unset SMCMI_NO_RNG_AHEAD
echo "=== c_abi_callback (energy sums in the accept launch)"
gcc -O2 -std=c99 -ffp-contract=off -fopenmp -DCB_THREADS=8 -I include -o examples/c_abi_callback examples/c_abi_callback.c -L smc.jl_amd/csrc -lsmcmi -lm -Wl,-rpath,$PWD/smc.jl_amd/csrc
for rep in 1 2 3; do OMP_WAIT_POLICY=ACTIVE OMP_PROC_BIND=close LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback 2>&1 | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['callback']; print(d['particle_stages_per_s'], d['calls'], d['ms_per_stage'], {k:round(v,4) for k,v in d['phases_ms_per_stage'].items()})"; done
echo "=== full GPU suite"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12
