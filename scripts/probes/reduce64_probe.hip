// reduce64_probe.hip -- what does the 64-lane transposing reduction of KP = 50 doubles per lane cost on an MI355X at ONE wave per SIMD
// (the fp64 lane kernels' regime), by scheme?  Each iteration forms 50 lane-partials (one fused multiply-add each, as the axpy of a
// gradient pass does per lane set) and reduces them so that every dimension's total ends in one lane.
//   mode 0: lane_eval.hpp's scheme (round 3): per column two v_permlane32_swap folds and one v_permlane16_swap fold (three
//           instructions per fold on doubles), the 16-lane sums left out (they go through LDS in the kernel)
//   mode 1: a transposing DPP butterfly inside each 16-lane row (reg_eval.hpp's fold_pair on doubles: four selects, two DPP moves, one
//           add per fold), sixteen partials at a time, then two swap-fold levels across the rows
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I poismf_amd/csrc scripts/probes/reduce64_probe.hip -o /tmp/r64 && /tmp/r64
#include <hip/hip_runtime.h>
#include <cstdio>
#define real_t double
#include "lane_eval.hpp"
using namespace pmf;

template <int MODE> __global__ __launch_bounds__(64) void k(double* out, int iters, double seed)
{
    const int lane = threadIdx.x & 63;
    double p[50];
    for (int i = 0; i < 50; i++) p[i] = seed * (i + 1) + lane;
    double acc = 0.0;
    const bool c8 = (lane & 8) != 0, c4 = (lane & 4) != 0, c2 = (lane & 2) != 0, c1 = (lane & 1) != 0;
    for (int it = 0; it < iters; it++) {
        double q50[50];
#pragma unroll
        for (int i = 0; i < 50; i++) q50[i] = __builtin_fma(p[i], seed, acc);
        double r;
        if (MODE == 0) {
            r = 0.0;
#pragma unroll
            for (int c = 0; c < 13; c++) {
                const double x = c + 26 < 50 ? swap_fold<32>(q50[c], q50[c + 26]) : swap_fold<32>(q50[c], q50[c]);
                const double y = c + 39 < 50 ? swap_fold<32>(q50[c + 13], q50[c + 39]) : swap_fold<32>(q50[c + 13], q50[c + 13]);
                r += swap_fold<16>(x, y);
            }
        } else {
            double rows[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int N = 50 - 16 * b < 16 ? 50 - 16 * b : 16;
                double q[8], rr[4], s2[2];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (i + 8 < N) q[i] = fold_pair<0x128>(c8, q50[16 * b + i], q50[16 * b + i + 8]);
                    else if (i < N) q[i] = fold_one<0x128>(q50[16 * b + i]);
                    else q[i] = 0.0;
                }
                if (N >= 8) fold_level<0x141, 4, 8>(c4, q, rr); else fold_level<0x141, 4, 2>(c4, q, rr);
                if (N >= 4) fold_level<0x4E, 2, 4>(c2, rr, s2); else fold_level<0x4E, 2, 2>(c2, rr, s2);
                rows[b] = fold_pair<0xB1>(c1, s2[0], s2[1]);
            }
            r = swap_fold<16>(swap_fold<32>(rows[0], rows[2]), swap_fold<32>(rows[1], rows[3]));
        }
        acc = r * 1e-300;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name, int waves_per_simd, double* out)
{
    const int iters = 2000, grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, 50, 1e-3);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, 1e-3);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %d wave(s)/SIMD: %.3f ms, %.1f ns per reduction of 50 doubles per wave\n", name, waves_per_simd, ms, ms * 1e6 / iters);
}
int main()
{
    double* out; hipMalloc(&out, 256 * 4 * 2 * 64 * sizeof(double));
    for (int w = 1; w <= 2; w++) { run<0>("swap folds (13 columns x 3)", w, out); run<1>("DPP butterfly + 3 swap folds", w, out); }
    return 0;
}
