// valu_probe.hip -- how long does one wave64 VALU instruction occupy a SIMD of an MI355X?  Plain v_fmac_f32, v_pk_fma_f32 and
// v_fma_f64 in eight independent chains, one and two waves per SIMD.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND> __global__ __launch_bounds__(64) void k(float* out, int iters, float seed)
{
    float a[8]; f2 p[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = (f2){ a[i], a[i] + 1 }; d[i] = a[i]; }
    const float m = seed * 0.5f; const f2 m2 = (f2){ m, m }; const double md = m;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, a[i]);
                else if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], m2, p[i]);
                else d[i] = __builtin_fma(d[i], md, d[i]);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, int waves_per_simd, float* out)
{
    const int iters = 20000, grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, 100, 1e-30f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, iters, 1e-30f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 64;   // per wave
    printf("%-14s %d wave(s)/SIMD: %.3f ms, %.2f ns per instruction per SIMD (x clock GHz = cycles)\n", name, waves_per_simd, ms,
           ms * 1e6 / (instr * waves_per_simd));
}
// dependent-issue latency: CH independent chains of v_fmac_f32, and v_permlane32_swap (one per two multiply-adds)
template <int CH, bool SWAP> __global__ __launch_bounds__(64) void kc(float* out, int iters, float seed)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
    const float m = seed * 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / CH; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) a[i] = __builtin_fmaf(a[i], m, a[i]);
            if (SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[0]), "+v"(a[1]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH, bool SWAP> void runc(int waves_per_simd, float* out)
{
    const int iters = 20000, grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kc<CH, SWAP>), dim3(grid), dim3(64), 0, 0, out, 100, 1e-30f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kc<CH, SWAP>), dim3(grid), dim3(64), 0, 0, out, iters, 1e-30f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("v_fmac_f32 in %d chain(s)%s, %d wave(s)/SIMD: %.2f ns per multiply-add per wave, %.2f per SIMD\n", CH, SWAP ? " + one permlane32_swap per group" : "",
           waves_per_simd, ms * 1e6 / ((double)iters * 64), ms * 1e6 / ((double)iters * 64 * waves_per_simd));
}
int main()
{
    float* out; hipMalloc(&out, 256 * 4 * 4 * 64 * sizeof(float));
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("clock rate attribute: %d kHz\n", clk);
    for (int w = 1; w <= 2; w++) { run<0>("v_fmac_f32", w, out); run<1>("v_pk_fma_f32", w, out); run<2>("v_fma_f64", w, out); }
    for (int w = 1; w <= 2; w++) { runc<1, false>(w, out); runc<2, false>(w, out); runc<4, false>(w, out); runc<8, false>(w, out); runc<4, true>(w, out); runc<8, true>(w, out); }
    return 0;
}
