// hipcc --offload-arch=gfx950 -O2 -o xcc_probe xcc_probe.hip && ./xcc_probe
// Which XCD does workgroup b of a 256-workgroup launch run on (s_getreg_b32 HW_REG_XCC_ID)?  The team kernels group workgroups
// by this id; correctness does not depend on the answer, same-XCD hand-offs are only faster.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main()
{
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    k<<<256, 256>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int cnt[16] = {0}; int match = 0;
    for (int b = 0; b < 256; b++) { cnt[h[b] & 15]++; match += (int)((h[b] & 7) == (unsigned)(b % 8)); }
    printf("raw id of blocks 0..9:"); for (int b = 0; b < 10; b++) printf(" %#x", h[b]); printf("\n");
    printf("blocks per (id & 15):"); for (int i = 0; i < 16; i++) printf(" %d", cnt[i]); printf("\nblocks with (id & 7) == b %% 8: %d of 256\n", match);
    return 0;
}
