// Experiment (development only): cost of the XCD-hierarchical software grid barrier of the CDNA4 guide (MI355X_MICROARCH.md,
// row barrier-xcd) in the geometry a persistent SMC stage kernel would have - 196 blocks x 512 threads, one per CU, every block
// publishing one 68-double row per round and totalling all rows after the barrier.  Every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/gridbar2 tools/exp/gridbar2.hip && tools/exp/gridbar2 [blocks] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bar {
    unsigned xcnt[8 * 32];     // per-XCC arrival counters, one 128-byte line each
    unsigned top[32];
    unsigned gen[8 * 32];      // per-XCC generation words
    unsigned census[8 * 32];
    unsigned census_done[32];
    unsigned fail[32];
};
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ inline unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }

// returns false on timeout.  n_here = blocks on this XCC, n_xcc = XCCs that hold blocks (both from the census).
__device__ inline bool barrier_xcd(Bar *b, unsigned round, unsigned xcc, unsigned n_here, unsigned n_xcc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(&b->xcnt[xcc * 32], 1u, RLX);
        long spins = 0;
        if (old + 1 == n_here * round) {                      // last arriver of this XCC: go to the top level
            __hip_atomic_fetch_add(&b->top[0], 1u, RLX);
            while (__hip_atomic_load(&b->top[0], RLX) < n_xcc * round) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000) { ok = false; break; }
            }
            __hip_atomic_store(&b->gen[xcc * 32], round, RLX);
        } else {
            while (__hip_atomic_load(&b->gen[xcc * 32], RLX) < round) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000) { ok = false; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!ok) __hip_atomic_store(&b->fail[0], 1u, RLX);
    }
    __syncthreads();
    return ok;
}

template <int M>
__global__ void __launch_bounds__(512) k_rounds(double *rows, Bar *b, int rounds, double *out, long long *ticks, int payload) {
    __shared__ unsigned s_n[2];
    __shared__ double s_part[8 * M];
    const unsigned xcc = xcc_id();
    // census: how many blocks sit on each XCC (placement is not a contract: measured, then used)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&b->census[xcc * 32], 1u, RLX);
        __hip_atomic_fetch_add(&b->census_done[0], 1u, RLX);
        long spins = 0;
        while (__hip_atomic_load(&b->census_done[0], RLX) < gridDim.x) { __builtin_amdgcn_s_sleep(1); if (++spins > 4000000) break; }
        unsigned nx = 0;
        for (int k = 0; k < 8; ++k) nx += __hip_atomic_load(&b->census[k * 32], RLX) > 0 ? 1u : 0u;
        s_n[0] = __hip_atomic_load(&b->census[xcc * 32], RLX);
        s_n[1] = nx;
    }
    __syncthreads();
    const unsigned n_here = s_n[0], n_xcc = s_n[1];
    double acc = 0.0;
    long long t0 = 0, t1 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) t0 = __builtin_readcyclecounter();
    for (int r = 1; r <= rounds; ++r) {
        double *mine = rows + ((size_t)(r & 1) * gridDim.x + blockIdx.x) * M;
        if (payload && threadIdx.x < M) __hip_atomic_store(&mine[threadIdx.x], (double)(r + (int)threadIdx.x) + acc * 1e-300, RLX);   // sc1 store
        if (!barrier_xcd(b, (unsigned)r, xcc, n_here, n_xcc)) break;
        if (payload) {
            // every block totals all rows: thread (slice s, column c) adds the rows of its slice, slices combined through LDS
            const int c = threadIdx.x % M, s = threadIdx.x / M, S = 512 / M;
            double a = 0.0;
            if (s < S) {
                const double *base = rows + (size_t)(r & 1) * gridDim.x * M + c;
                for (int q = s; q < (int)gridDim.x; q += S) a += base[(size_t)q * M];
                s_part[s * M + c] = a;
            }
            __syncthreads();
            if (threadIdx.x < M) { double t = 0.0; for (int q = 0; q < S; ++q) t += s_part[q * M + threadIdx.x]; acc += t; }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { t1 = __builtin_readcyclecounter(); ticks[0] = t1 - t0; }
    if (threadIdx.x < M) out[blockIdx.x * M + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 196, rounds = argc > 2 ? atoi(argv[2]) : 2000;
    constexpr int M = 64;
    double *rows, *out; Bar *bar; long long *ticks;
    CHECK(hipMalloc(&rows, sizeof(double) * 2 * nb * M)); CHECK(hipMalloc(&out, sizeof(double) * nb * M));
    CHECK(hipMalloc(&bar, sizeof(Bar))); CHECK(hipMalloc(&ticks, 64));
    for (int payload = 0; payload < 2; ++payload) {
        CHECK(hipMemset(bar, 0, sizeof(Bar))); CHECK(hipMemset(rows, 0, sizeof(double) * 2 * nb * M));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_rounds<M><<<nb, 512>>>(rows, bar, rounds, out, ticks, payload);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        Bar hb; CHECK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
        long long tk; CHECK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
        unsigned cs[8]; for (int k = 0; k < 8; ++k) cs[k] = hb.census[k * 32];
        printf("{\"blocks\": %d, \"rounds\": %d, \"payload_rows\": %d, \"us_per_round\": %.3f, \"fail\": %u, \"census\": [%u,%u,%u,%u,%u,%u,%u,%u]}\n",
               nb, rounds, payload, 1e3 * ms / rounds, hb.fail[0], cs[0], cs[1], cs[2], cs[3], cs[4], cs[5], cs[6], cs[7]);
    }
    return 0;
}
