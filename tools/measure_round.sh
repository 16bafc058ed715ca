#!/bin/bash
# End-of-round measurement pipeline (one MI355X): every file profiles/README.md lists, into gpurun_out/final/.
# usage (GPU box): bash tools/measure_round.sh [rNN]
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
python bench.py --steps 8 --warmup 2 2>/dev/null | tail -1 > $OUT/${R}_bench.json
python bench.py --nparts 1000000 --no-history --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_1e6.json
python bench.py --nparts 10000000 --no-history --no-cpu --steps 1 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_1e7.json
# configs 4 and 5 with their own cpu_baseline (the CPU leg runs each oracle variant once on the full workload)
python bench.py --workload capm --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_capm.json
python bench.py --workload kalman --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_kalman.json
# config 5 on its stated machine: 50 000 particles on 4 GPUs = 12 500 per GPU (lane-split filter), and the round-2 kernel on the same cloud
python bench.py --workload kalman --nparts 12500 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_kalman_n12500.json
SMCMI_ENGINE=1 python bench.py --workload kalman --nparts 12500 --no-cpu --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_kalman_n12500_engine1_stage.json
python bench.py --workload kalman --nparts 25000 --no-cpu --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_kalman_n25000.json
SMCMI_ENGINE=1 python bench.py --workload kalman --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_kalman_engine1_stage.json
gcc -O2 -std=c99 -ffp-contract=off -fopenmp -DCB_THREADS=8 -I include -o examples/c_abi_callback examples/c_abi_callback.c -L smc.jl_amd/csrc -lsmcmi -lm -Wl,-rpath,$ROOT/smc.jl_amd/csrc   # (against THIS build's struct layouts)
OMP_WAIT_POLICY=ACTIVE OMP_PROC_BIND=close LD_LIBRARY_PATH=smc.jl_amd/csrc:/opt/rocm/lib ./examples/c_abi_callback > $OUT/${R}_callback_c.json 2>/dev/null
# where a stage of the closure path spends its time: the batch at once / in chunks, the example's callback on 1 / 8 threads
bash tools/callback_phases.sh > $OUT/${R}_callback_phases.json 2>/dev/null
# FP64 matrix pipe beside the vector pipe (config 5's question: tools/ubench/mfma_f64.hip)
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -w -o mfma_f64 mfma_f64.hip > /dev/null 2>&1 && ./mfma_f64 > $OUT/${R}_mfma_f64.json 2>/dev/null)
python bench.py --alpha 0.9 --no-cpu --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_alpha09.json
# the reference's default schedule (fixed, 300 stages) at 100 000 / 5 000 / 1 000 particles: one hand-over per stage, and with exact shifts (two)
python tools/fixed_schedule.py > $OUT/${R}_fixed_schedule.txt 2>/dev/null
SMCMI_SHIFT_LAG=0 python tools/fixed_schedule.py 100000 5000 >> $OUT/${R}_fixed_schedule.txt 2>/dev/null
# clouds beyond 131 072 particles on one handle: two chunks per segment worker (adaptive: config 2's schedule; the default schedule)
python bench.py --nparts 250000 --no-history --no-cpu --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_n250000.json
python tools/fixed_schedule.py 250000 2>/dev/null | head -1 >> $OUT/${R}_fixed_schedule.txt
# ... and one rank's share of such a run on 8 GPUs (125 000 particles, sharded segments): riding / exact shifts
bash tools/fixed_schedule_shard.sh >> $OUT/${R}_fixed_schedule.txt 2>/dev/null
# the driver's RCCL branch as 8 ranks sharing this GPU (tests/fake_rccl): the multi-rank bench line with its pre-flight verdict
make -C tests/fake_rccl libfake_rccl.so > /dev/null 2>&1
HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_BENCH_COMM=rccl_shared SMCMI_RCCL_PATH=$ROOT/tests/fake_rccl/libfake_rccl.so python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29520 \
    bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu --no-history --nparts 65536 2>/dev/null | grep '^{' | tail -1 > $OUT/${R}_bench_8ranks_one_gpu_fake_rccl.json
python bench.py --alpha 0.9 --nparts 1000000 --no-history --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/${R}_bench_1e6_alpha09.json
# one rank's share of config 3 on 8 GPUs (125 000 particles, every hand-over through the all-gather path of a 1-rank RCCL communicator)
HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_FORCE_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu --no-history --nparts 125000 2>/dev/null | grep '^{' | tail -1 > $OUT/${R}_shard_rank_125k.json
# the same rank with the peer mailbox forced on (system-scope hand-overs): its stages run inside sharded segments
HSA_ENABLE_IPC_MODE_LEGACY=0 SMCMI_FORCE_SHARDED=1 SMCMI_MAILBOX=2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 \
    bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu --no-history --nparts 125000 2>/dev/null | grep '^{' | tail -1 > $OUT/${R}_shard_rank_125k_segments.json
# one rank's share of config 3 on 4 / 2 GPUs (250 000 / 500 000 particles: engine 2's large-shard stage, csrc/stage2b.hpp) with its kernel table
bash tools/shard_rank_prof.sh final_shard 250000 500000 > $OUT/${R}_shard_rank_large.log 2>&1
for n in 250000 500000; do
    cp gpurun_out/final_shard/bench_n${n}_mb2.json $OUT/${R}_shard_rank_${n}.json
    cp gpurun_out/final_shard/kernel_stats_n${n}.txt $OUT/${R}_kernel_stats_shard_rank_${n}.txt
done
cd /tmp && export TMPDIR=/tmp
pmc() {   # pmc <tag> <n> <bench args...>: kernel table + the three counter passes of one configuration
    local tag=$1 n=$2; shift 2
    rocprofv3 --kernel-trace --stats -d $OUT/kt_$tag -o kt -- python $ROOT/bench.py --no-cpu "$@" 2>/dev/null | tail -1 > $OUT/${R}_bench_${tag}_under_rocprof.json
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf_$tag -o f -- python $ROOT/bench.py --no-cpu "$@" > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw_$tag -o w -- python $ROOT/bench.py --no-cpu "$@" > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/ps_$tag -o s -- python $ROOT/bench.py --no-cpu "$@" > /dev/null 2>&1
    python $ROOT/profiles/summarize_rocpd.py $(find $OUT/kt_$tag -name "*.db" | head -1) > $OUT/${R}_kernel_stats_${tag}.txt
    python $ROOT/profiles/pmc_extract.py $(find $OUT/pf_$tag -name "*.db" | head -1) $(find $OUT/pw_$tag -name "*.db" | head -1) $n $(find $OUT/ps_$tag -name "*.db" | head -1) > $OUT/${R}_pmc_${tag}.json
    rm -rf $OUT/kt_$tag $OUT/pf_$tag $OUT/pw_$tag $OUT/ps_$tag
}
kt() {    # kt <tag> <bench args...>: kernel table only
    local tag=$1; shift
    rocprofv3 --kernel-trace --stats -d $OUT/kt_$tag -o kt -- python $ROOT/bench.py --no-cpu "$@" > /dev/null 2>&1
    python $ROOT/profiles/summarize_rocpd.py $(find $OUT/kt_$tag -name "*.db" | head -1) > $OUT/${R}_kernel_stats_${tag}.txt
    rm -rf $OUT/kt_$tag
}
pmc capm_n200000 200000 --workload capm --steps 2 --warmup 1
kt kalman --workload kalman --steps 2 --warmup 1
kt kalman_n12500 --workload kalman --nparts 12500 --steps 2 --warmup 1
kt gauss10_n1000000_alpha09 --alpha 0.9 --nparts 1000000 --no-history --steps 1 --warmup 1
pmc gauss10_n100000 100000 --steps 3 --warmup 1
kt gauss10_n100000_alpha09 --alpha 0.9 --steps 3 --warmup 1
pmc kalman_n50000 50000 --workload kalman --steps 2 --warmup 1
pmc kalman_n12500 12500 --workload kalman --nparts 12500 --steps 2 --warmup 1
pmc gauss10_n1000000 1000000 --nparts 1000000 --no-history --steps 1 --warmup 1
pmc gauss10_n10000000 10000000 --nparts 10000000 --no-history --steps 1 --warmup 0
ls -la $OUT
