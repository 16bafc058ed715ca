#!/bin/bash
# development: the phases of a stage that resamples inside a segment (SMCMI_PROF2) next to its neighbour
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
ST=$(python - <<'P'
import sys; sys.path.insert(0, '.')
import numpy as np
from smc_jl_amd import Engine
from smc_jl_amd.host import workloads
e = Engine(100000, 10, seed=1, max_stages=1500)
e.set_model(workloads.gauss_spec(10)); e.init_from_prior()
r = e.run(use_fixed_schedule=False, tempering_target=0.97)
rec = e.stage_records(r["n_stages"])
rs = [i + 1 for i, v in enumerate(rec["resampled"]) if v]
print(rs[3])
P
)
echo "resample stage (record index + 1): $ST"
for s in $ST $((ST+1)) $((ST+3)); do
  SMCMI_PROF2=$s python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | grep -E "smcmi3\] stage|selection inside" | tail -2
done
