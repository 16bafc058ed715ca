// Development probe: which die (XCC_ID) and CU does block b of a one-block-per-CU launch land on?
//   hipcc --offload-arch=gfx950 -O2 -w -o xcc_map xcc_map.hip && ./xcc_map [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(512) k(unsigned *out) {
    extern __shared__ double pad[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        out[2 * blockIdx.x] = xcc & 0xF;
        out[2 * blockIdx.x + 1] = hw;
        pad[0] = 1.0;
    }
    // keep the block alive a little so that all are resident together
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 208;
    unsigned *d, h[2 * 1024];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d, 0xFF, sizeof(h));
        k<<<grid, 512, 96 * 1024>>>(d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int match = 0;
        for (int b = 0; b < grid; ++b) match += (int)h[2 * b] == b % 8;
        printf("launch %d: %d of %d blocks on die (block %% 8); first 32 dies:", rep, match, grid);
        for (int b = 0; b < 32 && b < grid; ++b) printf(" %u", h[2 * b]);
        printf("\n");
    }
    return 0;
}
