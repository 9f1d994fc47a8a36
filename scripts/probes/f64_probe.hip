// f64_probe.hip -- what the pieces of a lane-engine fp64 evaluation cost on an MI355X at ONE wave per SIMD (the k = 100 kernels' occupancy):
// a dependent v_fmac_f64 chain (1 / 2 / 4 chains), the same with its multiplicand read out of AGPRs first (what hipcc does with the tile of the
// TNCG instances), v_accvgpr_read alone, and a double swap_fold.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/f64_probe.hip -o /tmp/f64_probe && /tmp/f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
// KIND 0: CH chains of v_fmac_f64 on VGPRs; 1: one chain, multiplicand through 2 v_accvgpr_read per step; 2: v_accvgpr_read only (2 per step);
// 3: one chain, multiplicand via v_accvgpr_read issued 4 steps AHEAD (software-pipelined); 4: double swap_fold<32> chain (2 swaps + add)
template <int KIND, int CH> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(double* out, int iters, double seed)
{
    double acc[4] = { seed, seed + 1, seed + 2, seed + 3 };
    double m[8];
    for (int i = 0; i < 8; i++) m[i] = seed * 1e-3 + i + threadIdx.x;
    unsigned alo[8], ahi[8];
    for (int i = 0; i < 8; i++) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, m[i]);
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(alo[i]) : "v"((unsigned)b));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ahi[i]) : "v"((unsigned)(b >> 32)));
    }
    const double x = seed * 0.5;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64; r++) {
            if constexpr (KIND == 0) {
                acc[r % CH] = __builtin_fma(m[r % 8], x, acc[r % CH]);
                asm volatile("" : "+v"(acc[r % CH]));
            } else if constexpr (KIND == 1) {
                unsigned lo, hi;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(alo[r % 8]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(ahi[r % 8]));
                const double t = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
                acc[r % CH] = __builtin_fma(t, x, acc[r % CH]);
                asm volatile("" : "+v"(acc[r % CH]));
            } else if constexpr (KIND == 2) {
                unsigned lo, hi;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(alo[r % 8]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(ahi[r % 8]));
                asm volatile("" :: "v"(lo), "v"(hi));
            } else if constexpr (KIND == 4) {
                const unsigned long long ab = __builtin_bit_cast(unsigned long long, acc[0]), bb = __builtin_bit_cast(unsigned long long, acc[1]);
                const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)ab, (unsigned)bb, false, false);
                const auto rh = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
                acc[0] = __builtin_bit_cast(double, ((unsigned long long)rh[0] << 32) | rl[0]) + __builtin_bit_cast(double, ((unsigned long long)rh[1] << 32) | rl[1]);
                asm volatile("" : "+v"(acc[0]));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int KIND, int CH> void run(const char* name, double* out)
{
    const int iters = 4000, grid = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, CH>), dim3(grid), dim3(64), 0, 0, out, 50, 1e-30);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, CH>), dim3(grid), dim3(64), 0, 0, out, iters, 1e-30);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %.2f ns per step (x ~2.3 GHz = %.1f cycles)\n", name, ms * 1e6 / ((double)iters * 64), ms * 1e6 / ((double)iters * 64) * 2.3);
}
int main()
{
    double* out; hipMalloc(&out, 256 * 4 * 64 * sizeof(double));
    run<0, 1>("v_fmac_f64, one dependent chain", out);
    run<0, 2>("v_fmac_f64, two chains", out);
    run<0, 4>("v_fmac_f64, four chains", out);
    run<1, 1>("2 x v_accvgpr_read + v_fmac_f64, one chain", out);
    run<1, 2>("2 x v_accvgpr_read + v_fmac_f64, two chains", out);
    run<1, 4>("2 x v_accvgpr_read + v_fmac_f64, four chains", out);
    run<2, 1>("2 x v_accvgpr_read alone", out);
    run<4, 1>("double swap_fold<32> (2 swaps + add), dependent", out);
    return 0;
}
