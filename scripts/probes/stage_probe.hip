// stage_probe.hip -- development probe: how fast can pageable host arrays reach the device through pinned chunks?  Sweeps the chunk
// size, the number of filling threads and the number of DMA queues (streams) of a pipeline shaped like devmem.hpp's
// pmf_upload_staged; fills are a memcpy (values, factors) or the size_t -> u32 narrowing (indices).  Also a device KERNEL reading
// the pinned chunks over PCIe instead of the DMA engine.   hipcc -O3 --offload-arch=gfx950 -o stage_probe stage_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void pull(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main()
{
    const size_t n = (size_t)100 << 20;   // elements: 400 MB of u32 out, 800 MB of size_t in
    std::vector<unsigned long long> wide(n);
    std::vector<unsigned> vals(n);
    for (size_t i = 0; i < n; i++) { wide[i] = i * 7 % 100000; vals[i] = (unsigned)i; }
    unsigned* d; hipMalloc(&d, n * 4);
    hipStream_t st[4]; for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int MAXT = 32;
    const size_t MAXCH = (size_t)16 << 20;
    std::vector<void*> pin(2 * MAXT); std::vector<hipEvent_t> ev(2 * MAXT);
    for (int i = 0; i < 2 * MAXT; i++) { hipHostMalloc(&pin[i], MAXCH, hipHostMallocDefault); memset(pin[i], 0, MAXCH); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
    for (int mode = 0; mode < 3; mode++)            // 0 memcpy fill, 1 narrowing fill, 2 narrowing fill + kernel pull
    for (size_t chunk : { (size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20 })
    for (int nt : { 8, 16, 32 })
    for (int nq : { 1, 2, 4 }) {
        if (mode == 2 && nq != 1) continue;
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            const double t0 = now();
            const size_t per = chunk / 4;
            auto work = [&](int t) {
                hipSetDevice(0);
                const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
                hipStream_t s = st[t % nq];
                int b = 0;
                for (size_t i = lo; i < hi; i += per, b ^= 1) {
                    const size_t cnt = hi - i < per ? hi - i : per;
                    const int slot = 2 * t + b;
                    hipEventSynchronize(ev[slot]);
                    unsigned* o = (unsigned*)pin[slot];
                    if (mode == 0) memcpy(o, vals.data() + i, cnt * 4);
                    else { const unsigned long long* q = wide.data() + i; for (size_t j = 0; j < cnt; j++) o[j] = (unsigned)q[j]; }
                    if (mode == 2) hipLaunchKernelGGL(pull, dim3(64), dim3(256), 0, s, (const uint4*)o, (uint4*)(d + i), cnt / 4);
                    else hipMemcpyAsync(d + i, o, cnt * 4, hipMemcpyHostToDevice, s);
                    hipEventRecord(ev[slot], s);
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(work, t);
            work(0);
            for (auto& x : th) x.join();
            const double t1 = now();
            for (int q = 0; q < nq; q++) hipStreamSynchronize(st[q]);
            const double t2 = now();
            if (t2 - t0 < best) best = t2 - t0;
            (void)t1;
        }
        printf("%s chunk %2zu MB threads %2d queues %d: %.2f ms = %.1f GB/s\n", mode == 0 ? "memcpy fill   " : mode == 1 ? "narrowing fill" : "narrow + kernel pull", chunk >> 20, nt, nq, best, n * 4 / best * 1e-6);
        fflush(stdout);
    }
    // the way back: device -> pinned -> pageable
    for (size_t chunk : { (size_t)4 << 20, (size_t)16 << 20 })
    for (int nt : { 8, 16 }) {
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            const double t0 = now();
            const size_t per = chunk / 4;
            auto work = [&](int t) {
                hipSetDevice(0);
                const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
                hipStream_t s = st[0];
                auto issue = [&](size_t i, int b) { const size_t cnt = hi - i < per ? hi - i : per; hipMemcpyAsync(pin[2 * t + b], d + i, cnt * 4, hipMemcpyDeviceToHost, s); hipEventRecord(ev[2 * t + b], s); };
                if (lo < hi) issue(lo, 0);
                int b = 0;
                for (size_t i = lo; i < hi; i += per, b ^= 1) {
                    if (i + per < hi) issue(i + per, b ^ 1);
                    hipEventSynchronize(ev[2 * t + b]);
                    memcpy(vals.data() + i, pin[2 * t + b], (hi - i < per ? hi - i : per) * 4);
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(work, t);
            work(0);
            for (auto& x : th) x.join();
            const double t2 = now();
            if (t2 - t0 < best) best = t2 - t0;
        }
        printf("download chunk %2zu MB threads %2d: %.2f ms = %.1f GB/s\n", chunk >> 20, nt, best, n * 4 / best * 1e-6);
    }
    return 0;
}
