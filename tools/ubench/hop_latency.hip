// Development probe: store -> load hop between two blocks, same die (XCD) vs different dies, agent-scope (sc1) loads vs loads answered by
// the die's L2 (sc0 on a line the reader's CU has never touched: every round uses a fresh 128-byte line).
//   hipcc --offload-arch=gfx950 -O2 -w -o hop_latency hop_latency.hip && ./hop_latency
// Two blocks play ping-pong `rounds` times: A stores word[r] = r+1 (sc1), B polls it and answers in its own array, A polls that.
// Reported: ns per one-way hop = wall time / (2 * rounds).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int LINE = 16;   // u64 per 128-byte line
template <int MODE>   // 0: sc1 loads, 1: sc0 loads
__device__ inline unsigned long long ld(const unsigned long long *p) {
    unsigned long long v;
    if (MODE == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ inline void st(unsigned long long *p, unsigned long long v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int MODE>
__global__ void __launch_bounds__(64) k(unsigned long long *ping, unsigned long long *pong, int rounds, int blk_a, int blk_b, long long *out, int spin_lines) {
    extern __shared__ double pad[];
    if (threadIdx.x != 0) return;
    pad[0] = 0;
    const bool a = (int)blockIdx.x == blk_a, b = (int)blockIdx.x == blk_b;
    if (!a && !b) return;
    long long t0 = wall_clock64();
    long long bad = 0;
    for (int r = 0; r < rounds; ++r) {
        unsigned long long *pi = ping + (long long)r * LINE, *po = pong + (long long)r * LINE;
        if (a) {
            st(pi, (unsigned long long)(r + 1));
            // sc0 mode: a line this CU has never read - ONE probe per line would be the honest L2 read; polling needs re-reads, which may
            // hit the vector cache: walk over `spin_lines` alias copies?  no: simply re-issue and count time-outs
            long long n = 0;
            while (ld<MODE>(po) != (unsigned long long)(r + 1)) { if (++n > 100000) { ++bad; break; } }
        } else {
            long long n = 0;
            while (ld<MODE>(pi) != (unsigned long long)(r + 1)) { if (++n > 100000) { ++bad; break; } }
            st(po, (unsigned long long)(r + 1));
        }
    }
    long long t1 = wall_clock64();
    if (a) { out[0] = t1 - t0; out[1] = bad; } else out[2] = bad;
}
template <int MODE>
static void run(const char *name, int blk_b, unsigned long long *ping, unsigned long long *pong, long long *d_out, int rounds) {
    hipMemset(ping, 0, (size_t)rounds * LINE * 8); hipMemset(pong, 0, (size_t)rounds * LINE * 8);
    hipMemset(d_out, 0, 64);
    hipDeviceSynchronize();
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    k<MODE><<<64, 64, 96 * 1024>>>(ping, pong, rounds, 0, blk_b, d_out, 0);
    long long h[3];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("  {\"case\": \"%s\", \"partner_block\": %d, \"ns_per_hop\": %.1f, \"timed_out_a\": %lld, \"timed_out_b\": %lld},\n", name, blk_b, h[0] * 10.0 / (2.0 * rounds), h[1], h[2]);
}
int main() {
    const int rounds = 2000;
    unsigned long long *ping, *pong; long long *d_out;
    hipMalloc(&ping, (size_t)rounds * LINE * 8); hipMalloc(&pong, (size_t)rounds * LINE * 8); hipMalloc(&d_out, 64);
    printf("{\"rows\": [\n");
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("sc1 loads, same die (block 8)", 8, ping, pong, d_out, rounds);
        run<0>("sc1 loads, other die (block 1)", 1, ping, pong, d_out, rounds);
        run<2>("sc0 sc1 loads, same die", 8, ping, pong, d_out, rounds);
        run<1>("sc0 loads, same die (block 8)", 8, ping, pong, d_out, rounds);
        run<1>("sc0 loads, other die (block 1)", 1, ping, pong, d_out, rounds);
    }
    printf("  {}]}\n");
    return 0;
}
