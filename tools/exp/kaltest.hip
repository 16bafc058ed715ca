// development: the Kalman filter alone, to read its instruction mix quickly
#include <hip/hip_runtime.h>
#include <cmath>
#include "../../include/smcmi.h"
#include "../../smc.jl_amd/csrc/devstate.hpp"
#include "../../smc.jl_amd/csrc/model.hpp"
namespace smcmi {
__global__ void __launch_bounds__(256) k_kal(const double *th, const double *y, long long nt, long long mid, const double *aux, double *out) {
    double t[13];
    for (int k = 0; k < 13; ++k) t[k] = th[k * 64 + threadIdx.x];
    const KalmanLL r = kalman_lgss2(t, y, nt, mid, aux, 0.2);
    out[threadIdx.x] = r.ll + r.ll_mid;
}
}
