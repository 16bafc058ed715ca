// Development microbenchmark: does the FP64 matrix pipe of gfx950 add to the FP64 vector rate that bounds the
// Kalman filters of config 5, or share it?
//
//   hipcc --offload-arch=gfx950 -O3 -w -o mfma_f64 mfma_f64.hip && ./mfma_f64       (prints one JSON object)
//
// One 512-thread block per CU (96 KB of dynamic LDS keeps a second block off the CU): wavefronts w and w + 4 of a block share a SIMD.
// Roles:  waves 0..3 run stream A, waves 4..7 run stream B (or nothing).  Every stream is `iters` trips of 64 instructions on 16
// independent accumulator sets (no dependent chain shorter than 16 instructions).  Time = HIP events around the launch (the launch
// overhead is measured with iters = 0 and subtracted).  Streams:
//   fma   : v_fma_f64                     128 flop per wavefront instruction
//   m16   : v_mfma_f64_16x16x4_f64       2 048 flop  (D[16x16] += A[16x4] B[4x16])
//   m4    : v_mfma_f64_4x4x4_4b_f64        512 flop  (four independent 4x4x4 blocks)
// Reported per configuration: ns per instruction and wavefront, TFLOP/s over the chip (1 024 SIMDs), and for the mixed runs the time
// against the two streams run alone: overlap = (t_A + t_B - t_AB) / min(t_A, t_B) - 1.0: the pipes run beside each other, 0.0: they share
// the issue slots / the datapath.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

enum { S_NONE = 0, S_FMA = 1, S_M16 = 2, S_M4 = 3 };

template <int S>
__device__ inline void stream(double *out, int iters, int slot) {
    const double a = out[threadIdx.x & 63], b = out[64 + (threadIdx.x & 63)];
    if constexpr (S == S_FMA) {
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = out[128 + i];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i];
        out[256 + slot] = s;
    } else if constexpr (S == S_M16) {
        double4_t acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = double4_t{out[128 + i], 0.0, 0.0, 0.0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        out[256 + slot] = s;
    } else if constexpr (S == S_M4) {
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = out[128 + i];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i];
        out[256 + slot] = s;
    }
}

template <int SA, int SB>
__global__ void __launch_bounds__(512) k(double *out, int iters) {
    extern __shared__ double pad[];
    if (threadIdx.x == 0 && iters < 0) pad[0] = 1.0;
    const int slot = blockIdx.x * 512 + threadIdx.x;
    if (threadIdx.x < 256) stream<SA>(out, iters, slot);
    else stream<SB>(out, iters, slot);
}

static double *d_out;
template <int SA, int SB>
static double time_ms(int iters, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 96 * 1024;
    hipFuncSetAttribute((const void *)k<SA, SB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k<SA, SB><<<grid, 512, lds>>>(d_out, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount, grid = n_cu;
    hipMalloc(&d_out, (size_t)(256 + 512 * 1024) * sizeof(double));
    hipMemset(d_out, 0, (size_t)(256 + 512 * 1024) * sizeof(double));
    const int iters = 4000;                          // x 64 instructions per trip
    const double n_inst = 64.0 * iters;
    const double t0 = time_ms<S_NONE, S_NONE>(0, grid);
    struct Row { const char *name; double ms; double flop_a, flop_b; };
    std::vector<Row> rows;
    auto add = [&](const char *name, double ms, double fa, double fb) { rows.push_back({name, ms - t0, fa, fb}); };
    add("fma_alone", time_ms<S_FMA, S_NONE>(iters, grid), 128, 0);
    add("fma_two_waves", time_ms<S_FMA, S_FMA>(iters, grid), 128, 128);
    add("m16_alone", time_ms<S_M16, S_NONE>(iters, grid), 2048, 0);
    add("m16_two_waves", time_ms<S_M16, S_M16>(iters, grid), 2048, 2048);
    add("m4_alone", time_ms<S_M4, S_NONE>(iters, grid), 512, 0);
    add("m4_two_waves", time_ms<S_M4, S_M4>(iters, grid), 512, 512);
    add("m16_beside_fma", time_ms<S_M16, S_FMA>(iters, grid), 2048, 128);
    add("m4_beside_fma", time_ms<S_M4, S_FMA>(iters, grid), 512, 128);
    const double simds = 4.0 * n_cu;
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, \"instructions_per_stream\": %.0f, \"launch_overhead_ms\": %.4f,\n \"rows\": [\n",
           p.name, p.gcnArchName, n_cu, p.clockRate / 1000, n_inst, t0);
    for (size_t i = 0; i < rows.size(); ++i) {
        const Row &r = rows[i];
        const double ns = r.ms * 1e6 / n_inst;
        const double tf = (r.flop_a + r.flop_b) * n_inst * simds / (r.ms * 1e-3) / 1e12;
        printf("  {\"config\": \"%s\", \"ms\": %.4f, \"ns_per_instruction_per_wavefront\": %.3f, \"tflops_chip\": %.2f}%s\n", r.name, r.ms, ns, tf, i + 1 < rows.size() ? "," : "");
    }
    auto get = [&](const char *n) { for (auto &r : rows) if (!strcmp(r.name, n)) return r.ms; return 0.0; };
    const double ov16 = (get("m16_alone") + get("fma_alone") - get("m16_beside_fma")) / std::min(get("m16_alone"), get("fma_alone"));
    const double ov4 = (get("m4_alone") + get("fma_alone") - get("m4_beside_fma")) / std::min(get("m4_alone"), get("fma_alone"));
    printf(" ],\n \"overlap_m16_with_fma\": %.3f, \"overlap_m4_with_fma\": %.3f,\n", ov16, ov4);
    printf(" \"note\": \"overlap 1 = the matrix pipe runs beside the vector pipe (the mixed run takes as long as the longer stream alone), 0 = they share (the mixed run takes the sum)\"}\n");
    return 0;
}
