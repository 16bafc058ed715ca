// Development microbenchmark: issue cost of the instructions the lane-split Kalman filter is made of (one wavefront per SIMD,
// 16 independent accumulators, 256 instructions per loop trip).   hipcc --offload-arch=gfx950 -O3 -o dpp_rate dpp_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, long long *clk, int iters) {
    double a[16], c = out[threadIdx.x & 15], x = out[16 + threadIdx.x];
    for (int i = 0; i < 16; ++i) a[i] = out[32 + i];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(x));
                if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c), "v"(x));
                if (MODE == 2) { int lo = __double2loint(a[i]); asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(lo) : "v"(lo)); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
                if (MODE == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(x));
            }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[64 + blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    double *o; long long *c; hipMalloc(&o, 4 << 20); hipMemset(o, 0, 4 << 20); hipMalloc(&c, 8 * 4096);
    const char *names[] = {"v_fmac_f64", "v_fmac_f64_dpp row_newbcast", "v_mov_b32_dpp quad_perm", "v_add_f64", "v_mul_f64", "v_fma_f64 (vop3)"};
    for (int waves = 1; waves <= 2; ++waves)
    for (int m = 0; m < 6; ++m) {
        const int iters = 200, grid = 256 * 4 * waves;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (m) { case 0: k<0><<<grid, 64>>>(o, c, iters); break; case 1: k<1><<<grid, 64>>>(o, c, iters); break; case 2: k<2><<<grid, 64>>>(o, c, iters); break;
                         case 3: k<3><<<grid, 64>>>(o, c, iters); break; case 4: k<4><<<grid, 64>>>(o, c, iters); break; default: k<5><<<grid, 64>>>(o, c, iters); }
            hipEventRecord(e1);
            hipDeviceSynchronize();
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        printf("waves/SIMD %d  %-30s %.2f shader-clock ticks per instruction (s_memtime units), %.3f ns per instruction per wavefront (kernel wall time)\n", waves, names[m], (double)h[0] / (iters * 256.0), ms * 1e6 / (iters * 256.0));
    }
    return 0;
}
