// stage_probe2.hip -- development probe: why do two DMA queues help stage_probe.hip's loop and not run_poismf's set-up?  The same
// pipeline with, one at a time: the event fences between the two queues, a device buffer allocated per upload, two uploads back to back.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const int NT = 8;
static const size_t CH = (size_t)4 << 20;
static std::vector<void*> pin(2 * NT);
static std::vector<hipEvent_t> ev(2 * NT);
static hipStream_t st[2];
static hipEvent_t evb, eve;
template <class Fill> static void staged(unsigned* d, size_t n, int nq, bool fences, Fill fill)
{
    if (fences && nq == 2) { hipEventRecord(evb, st[0]); hipStreamWaitEvent(st[1], evb, 0); }
    const size_t per = CH / 4;
    auto work = [&](int t) {
        hipSetDevice(0);
        const size_t lo = n * t / NT, hi = n * (t + 1) / NT;
        hipStream_t s = st[t % nq];
        int b = 0;
        for (size_t i = lo; i < hi; i += per, b ^= 1) {
            const size_t cnt = hi - i < per ? hi - i : per;
            const int slot = 2 * t + b;
            hipEventSynchronize(ev[slot]);
            fill((unsigned*)pin[slot], i, cnt);
            hipMemcpyAsync(d + i, pin[slot], cnt * 4, hipMemcpyHostToDevice, s);
            hipEventRecord(ev[slot], s);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < NT; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    if (fences && nq == 2) { hipEventRecord(eve, st[1]); hipStreamWaitEvent(st[0], eve, 0); }
}
int main()
{
    const size_t n = (size_t)100 << 20;
    std::vector<unsigned long long> wide(n);
    std::vector<unsigned> vals(n);
    for (size_t i = 0; i < n; i++) { wide[i] = i * 7 % 100000; vals[i] = (unsigned)i; }
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEventCreateWithFlags(&evb, hipEventDisableTiming); hipEventCreateWithFlags(&eve, hipEventDisableTiming);
    for (int i = 0; i < 2 * NT; i++) { hipHostMalloc(&pin[i], CH, hipHostMallocDefault); memset(pin[i], 0, CH); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
    auto narrow = [&](unsigned* o, size_t i, size_t cnt) { const unsigned long long* q = wide.data() + i; for (size_t j = 0; j < cnt; j++) o[j] = (unsigned)q[j]; };
    auto copy = [&](unsigned* o, size_t i, size_t cnt) { memcpy(o, vals.data() + i, cnt * 4); };
    unsigned *d0, *d1; hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4);
    for (int variant = 0; variant < 5; variant++)
    for (int nq = 1; nq <= 2; nq++) {
        for (int rep = 0; rep < 4; rep++) {
            unsigned *a = d0, *b = d1;
            const bool fresh = variant == 2 || variant == 4, fences = variant >= 1, memset_first = variant == 4;
            double t0 = now();
            if (fresh) { hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); }
            if (memset_first) { hipMemsetAsync(a, 0, n * 4, st[0]); hipMemsetAsync(b, 0, n * 4, st[0]); }
            const double ta = now();
            staged(a, n, nq, fences, narrow);
            const double t1 = now();
            if (variant >= 3) staged(b, n, nq, fences, copy);
            const double t2 = now();
            hipStreamSynchronize(st[0]); hipStreamSynchronize(st[1]);
            const double t3 = now();
            printf("variant %d (%s%s%s%s) queues %d rep %d: alloc %.2f, narrowing upload %.2f, %s%.2f, drain %.2f ms\n", variant, fences ? "fences" : "plain", fresh ? ", fresh buffers" : "",
                   variant >= 3 ? ", two uploads" : "", memset_first ? ", memsets first" : "", nq, rep, ta - t0, t1 - ta, variant >= 3 ? "memcpy upload " : "-", t2 - t1, t3 - t2);
            if (fresh) { hipFree(a); hipFree(b); }
        }
    }
    return 0;
}
