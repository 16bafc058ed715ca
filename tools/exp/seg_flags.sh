#!/bin/bash
# development: compile the D = 10 segment kernel under candidate flag sets and print its resource line (no GPU needed)
cd "$(dirname "$0")/../../smc.jl_amd/csrc" && mkdir -p build_exp
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-value"
i=0
while IFS= read -r fl; do
  i=$((i+1))
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $fl -DSMCMI_INST3_D=10 -Rpass-analysis=kernel-resource-usage -c -o build_exp/e$i.o inst3.hip 2> build_exp/e$i.res
    echo "[$fl]"; grep -A12 "Function Name: _ZN5smcmi10k3_segmentILi10ELb1" build_exp/e$i.res | grep -E "VGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill" | sed 's/.*remark: [^ ]* *//' | tr '\n' ' '; echo ) > build_exp/e$i.out 2>&1 &
done <<'LIST'
-mllvm -sink-insts-to-avoid-spills
-mllvm -sink-insts-to-avoid-spills -mllvm -amdgpu-schedule-metric-bias=100
-mllvm -sink-insts-to-avoid-spills -mllvm -enable-misched=false
-mllvm -sink-insts-to-avoid-spills -mllvm -amdgpu-use-amdgpu-trackers=1
-mllvm -sink-insts-to-avoid-spills -mllvm -greedy-regclass-priority-trumps-globalness=1
-mllvm -sink-insts-to-avoid-spills -mllvm -disable-licm-promotion
-mllvm -sink-insts-to-avoid-spills -mllvm -disable-machine-licm
-mllvm -sink-insts-to-avoid-spills -mllvm -enable-post-misched=false
-mllvm -sink-insts-to-avoid-spills -mllvm -amdgpu-disable-unclustered-high-rp-reschedule
-mllvm -sink-insts-to-avoid-spills -mllvm -split-spill-mode=size
-mllvm -sink-insts-to-avoid-spills -mllvm -enable-gvn-hoist=false -mllvm -enable-loop-simplifycfg-term-folding
-mllvm -sink-insts-to-avoid-spills -mllvm -amdgpu-spill-vgpr-to-agpr=1
LIST
wait
cat build_exp/e*.out
