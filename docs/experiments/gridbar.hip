// Experiment: cost of a software grid barrier + exchange of per-block partial rows on MI355X (development only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ inline bool grid_barrier(unsigned *counter, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000) { ok = false; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

// flag barrier: every block publishes its round number in its own slot (no contended atomics); wave 0 of block 0 polls all
// slots with one coalesced load per lane and then publishes the release word that the other blocks poll.
__device__ inline bool flag_barrier(unsigned *flags /* [nb] */, unsigned *release, unsigned round) {
    __syncthreads();
    bool ok = true;
    const int nb = gridDim.x;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) {
            __threadfence();
            __hip_atomic_store(&flags[blockIdx.x], round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (blockIdx.x == 0) {
            long spins = 0;
            for (;;) {
                bool all = true;
                for (int b = threadIdx.x; b < nb; b += 64)
                    if (__hip_atomic_load(&flags[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round) all = false;
                if (__all(all)) break;
                if (++spins > 2000000) { ok = false; break; }
            }
            if (threadIdx.x == 0) __hip_atomic_store(release, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (threadIdx.x == 0) {
            long spins = 0;
            while (__hip_atomic_load(release, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000) { ok = false; break; }
            }
        }
        if (threadIdx.x == 0) __threadfence();
    }
    __syncthreads();
    return ok;
}

// each round: every block writes a row of M doubles, barrier, every block sums all rows (column c by thread c)
template <int M>
__global__ void k_rounds(double *rows, unsigned *counter, int rounds, double *out, long long *ticks, int mode, unsigned *flags) {
    const int nb = gridDim.x;
    double acc = 0.0;
    long long t0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        double *row = rows + ((size_t)(r & 1) * nb + blockIdx.x) * M;
        if (threadIdx.x < M) row[threadIdx.x] = (double)(blockIdx.x + threadIdx.x + r);
        if (!(mode >= 3 ? flag_barrier(flags, counter, (unsigned)(r + 1)) : grid_barrier(counter, (unsigned)(nb * (r + 1))))) { if (threadIdx.x == 0) out[blockIdx.x] = -1.0; return; }
        // all blocks read all rows: thread (s, c) adds rows s, s+S, ...
        const double *base = rows + (size_t)(r & 1) * nb * M;
        if (mode == 1 || mode == 4) {
            // thread (slice, c): S = T / M slices, rows slice, slice + S, ... all loads in flight, then LDS combine
            __shared__ double sc[1024];
            const int c = threadIdx.x % M, sl = threadIdx.x / M, S = blockDim.x / M;
            double a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { const int b = sl + q * S; a[q] = b < nb ? base[(size_t)b * M + c] : 0.0; }
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += a[q];
            sc[threadIdx.x] = s;
            __syncthreads();
            if (threadIdx.x < M) { double t = 0.0; for (int q = 0; q < S; ++q) t += sc[q * M + threadIdx.x]; acc += t; }
            __syncthreads();
        } else {
            if (threadIdx.x < M) acc += base[(size_t)blockIdx.x * M + threadIdx.x];   // barrier only (own row)
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = wall_clock64() - t0;
    if (threadIdx.x < M) out[(size_t)blockIdx.x * M + threadIdx.x] = acc;
}

int main() {
    const int nb = 256, T = 512, M = 32, rounds = 2000;
    double *rows, *out; unsigned *counter; long long *ticks;
    CHECK(hipMalloc(&rows, sizeof(double) * 2 * nb * M));
    CHECK(hipMalloc(&out, sizeof(double) * nb * M));
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&ticks, 8));
    unsigned *flags; CHECK(hipMalloc(&flags, 4 * nb));
    for (int rep = 0; rep < 4; ++rep) {
        int mode = 1 + rep;
        CHECK(hipMemset(flags, 0, 4 * nb));
        CHECK(hipMemset(counter, 0, 4));
        void *args[] = {&rows, &counter, (void *)&rounds, &out, &ticks, &mode, &flags};
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        CHECK(hipLaunchCooperativeKernel((void *)k_rounds<M>, dim3(nb), dim3(T), args, 0, 0));
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<double> h(nb * M);
        CHECK(hipMemcpy(h.data(), out, sizeof(double) * nb * M, hipMemcpyDeviceToHost));
        long long tk; CHECK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
        // expected: sum_r sum_b (b + c + r)
        double exp0 = 0; for (int r = 0; r < rounds; ++r) for (int b = 0; b < nb; ++b) exp0 += b + 0 + r;
        printf("mode %d rep %d: %.3f ms total, %.3f us per round (barrier + %d x %d-double exchange), check %s (%.0f vs %.0f), ticks/round %.1f\n", mode, rep, ms,
               1e3 * ms / rounds, nb, M, h[0] == exp0 && h[(nb - 1) * M] == exp0 ? "ok" : "MISMATCH", h[0], exp0, (double)tk / rounds);
    }
    return 0;
}
