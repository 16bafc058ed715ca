// development: compile single instantiations of the mutation kernels to read their register / scratch figures quickly
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c tools/exp/regtest.hip -o /dev/null
#include <hip/hip_runtime.h>
#include <cmath>
#include "../../include/smcmi.h"
#include "../../smc.jl_amd/csrc/devstate.hpp"
#include "../../smc.jl_amd/csrc/kernels.hpp"
#ifndef RT_D
#define RT_D 10
#endif
namespace smcmi {
template __global__ void k_mutate_reg<RT_D, false>(CloudPtrs, const DevState *, const ModelDev *, MutArgs, double *, int, int, int);
template __global__ void k_mutate_reg<RT_D, true>(CloudPtrs, const DevState *, const ModelDev *, MutArgs, double *, int, int, int);
}
