// mall_probe.hip -- development probe: random-row gather bandwidth (400-byte rows, the fp64 k = 50 factor row) as a
// function of the working-set size, to see what the 256 MiB Infinity Cache gives a re-gathering kernel.
//   hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip && ./mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

struct __attribute__((packed, aligned(8))) U16 { double v[2]; };

// one wave per "row of X": reads `per` factor rows (rowbytes each, 16-byte slots over 25 lanes x ...) named by idx
__global__ __launch_bounds__(256) void gather_kernel(const char* F, const unsigned* idx, size_t n_idx, int rowbytes, double* sink)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int slots = rowbytes / 16;            // 25
    const int rows_per_instr = 64 / slots;      // 2 (50 lanes busy)
    const int r_in = lane / slots, s_in = lane % slots;
    double acc = 0;
    // each wave handles chunks of 32 indices: 16 loads in flight
    for (size_t base = wave * 32; base + 32 <= n_idx; base += nwaves * 32) {
        U16 v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const unsigned c = idx[base + u * rows_per_instr + (r_in < rows_per_instr ? r_in : 0)];
            v[u] = *(const U16*)(F + (size_t)c * rowbytes + (size_t)(r_in < rows_per_instr ? s_in : 0) * 16);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) acc += v[u].v[0] + v[u].v[1];
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main()
{
    const int rowbytes = 400;
    const size_t max_rows = (size_t)2200 * 1000 * 1000 / rowbytes;
    char* F; hipMalloc(&F, max_rows * rowbytes + 64); hipMemset(F, 0, max_rows * rowbytes + 64);
    const size_t n_idx = (size_t)64 * 1000 * 1000;   // 64M gathered rows = 25.6 GB per launch
    unsigned* d_idx; hipMalloc(&d_idx, n_idx * 4);
    double* sink; hipMalloc(&sink, 8);
    std::vector<unsigned> h(n_idx);
    std::mt19937_64 rng(1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (double mb : { 25.0, 50.0, 100.0, 150.0, 200.0, 300.0, 400.0, 800.0, 2000.0 }) {
        const size_t rows = (size_t)(mb * 1e6 / rowbytes);
        for (size_t i = 0; i < n_idx; i++) h[i] = (unsigned)(rng() % rows);
        hipMemcpy(d_idx, h.data(), n_idx * 4, hipMemcpyHostToDevice);
        for (int grid : { 256 * 2, 256 * 8 }) {
            gather_kernel<<<grid, 256>>>(F, d_idx, n_idx, rowbytes, sink);
            hipEventRecord(e0);
            for (int it = 0; it < 3; it++) gather_kernel<<<grid, 256>>>(F, d_idx, n_idx, rowbytes, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("working set %7.0f MB  grid %5d x256: %.2f TB/s gathered\n", mb, grid, 3.0 * n_idx * rowbytes / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
