// spill_divergent.hip -- does a VGPR spill placed inside a lane-divergent region lose the inactive lanes' values?  (DESIGN.md 4.8's hazard.)
//
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/spill_divergent.hip -o /tmp/spill_divergent && /tmp/spill_divergent
//   hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S scripts/probes/spill_divergent.hip -o - | less     (the ISA)
//
// Construction: KEEP values per lane that are live in EVERY lane across a divergent region (`if (lane & 1)`), inside which HOT more values are
// made live at once, so that the allocator -- held to 128 registers by amdgpu_waves_per_eu(4,4) -- must spill some of the outer values INSIDE
// the region, where only the odd lanes are active.  After the lanes reconverge, every lane checks every outer value.  Three variants:
//   plain   the check reads each lane's own copy (what ordinary per-lane code does),
//   cross   the check reads the NEIGHBOUR lane's copy through a DPP row_shr -- a cross-lane read after reconvergence of a register that was
//           reloaded while its lane was switched off is exactly the situation the team kernels' exchanges are in,
//   asmdiv  the region's mask is narrowed by inline asm the compiler cannot see (s_and_saveexec around the hot block): the case where the
//           compiler's per-thread reasoning does not apply at all.
// Prints mismatches per variant; the ISA excerpt of the build this was last run with is kept next to it (spill_divergent.isa.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int KEEP = 96, HOT = 96;
__device__ __forceinline__ float hot_block(float seed, const float* __restrict__ tab)
{
    float h[HOT];
#pragma unroll
    for (int i = 0; i < HOT; i++) h[i] = tab[i] * seed + (float)i;
#pragma unroll
    for (int i = 0; i < HOT; i++) asm volatile("" : "+v"(h[i]));          // all HOT values live at once
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HOT; i++) s = __builtin_fmaf(h[i], h[(i * 7 + 3) % HOT], s);
    return s;
}
template <int VARIANT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe(const float* tab, unsigned* bad, float* sink)
{
    const int lane = threadIdx.x;
    float keep[KEEP];
#pragma unroll
    for (int i = 0; i < KEEP; i++) keep[i] = tab[i] + (float)(lane * 1000 + i);
#pragma unroll
    for (int i = 0; i < KEEP; i++) asm volatile("" : "+v"(keep[i]));       // defined in ALL lanes, before the region
    float s = 0.f;
    if constexpr (VARIANT == 2) {
        unsigned long long saved;
        const unsigned long long odd = 0xaaaaaaaaaaaaaaaaull;
        asm volatile("s_and_saveexec_b64 %0, %1" : "=s"(saved) : "s"(odd) : "memory");
        s = hot_block((float)lane, tab);
        asm volatile("s_mov_b64 exec, %0" :: "s"(saved) : "memory");
    } else if (lane & 1) s = hot_block((float)lane, tab);
    unsigned wrong = 0;
#pragma unroll
    for (int i = 0; i < KEEP; i++) {
        asm volatile("" : "+v"(keep[i]));
        float v = keep[i];
        int src = lane;
        if constexpr (VARIANT == 1) {                                      // the neighbour's copy (row_shr:1; lane 0 of a row keeps its own)
            v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
            src = (lane & 15) ? lane - 1 : lane;
        }
        wrong += v != tab[i] + (float)(src * 1000 + i);
    }
    if (wrong) atomicAdd(bad, wrong);
    sink[blockIdx.x * 64 + lane] = s;
}
int main()
{
    std::vector<float> tab(128);
    for (int i = 0; i < 128; i++) tab[i] = 0.5f + i;
    float *d_tab, *d_sink; unsigned* d_bad;
    (void)hipMalloc(&d_tab, 128 * 4); (void)hipMalloc(&d_sink, 1024 * 64 * 4); (void)hipMalloc(&d_bad, 12);
    (void)hipMemcpy(d_tab, tab.data(), 128 * 4, hipMemcpyHostToDevice);
    (void)hipMemset(d_bad, 0, 12);
    hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(64), 0, 0, d_tab, d_bad + 0, d_sink);
    hipLaunchKernelGGL(probe<1>, dim3(1024), dim3(64), 0, 0, d_tab, d_bad + 1, d_sink);
    hipLaunchKernelGGL(probe<2>, dim3(1024), dim3(64), 0, 0, d_tab, d_bad + 2, d_sink);
    unsigned bad[3];
    if (hipMemcpy(bad, d_bad, 12, hipMemcpyDeviceToHost) != hipSuccess) { printf("spill_divergent: no GPU\n"); return 2; }
    printf("spill_divergent: values of all-lane registers lost across a divergent region -- plain %u, cross-lane read %u, asm-narrowed EXEC %u (of %d)\n",
           bad[0], bad[1], bad[2], 1024 * 64 * KEEP);
    return 0;
}
