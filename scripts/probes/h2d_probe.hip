// h2d_probe.hip -- development probe: host -> device copy rates for pageable, registered and pinned memory, and the cost
// of hipMalloc / hipFree / hipHostRegister, to decide how run_poismf should upload caller-owned arrays.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 80u << 20;
    char* h = (char*)malloc(n); memset(h, 1, n);
    char* d; hipMalloc(&d, n);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 3; rep++) {
        double t = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); printf("pageable hipMemcpy 80 MiB: %.2f ms\n", now() - t);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipHostRegister(h, n, hipHostRegisterDefault); double t1 = now();
        hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t2 = now();
        hipHostUnregister(h); double t3 = now();
        printf("register %.2f ms, copy %.2f ms, unregister %.2f ms\n", t1 - t, t2 - t1, t3 - t2);
    }
    char* p; hipHostMalloc(&p, n); memset(p, 1, n);
    for (int rep = 0; rep < 2; rep++) { double t = now(); hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("pinned copy 80 MiB: %.2f ms\n", now() - t); }
    { double t = now(); memcpy(p, h, n); printf("host memcpy 80 MiB into pinned: %.2f ms\n", now() - t); }
    { double t = now(); hipMemcpy(h, d, n, hipMemcpyDeviceToHost); printf("pageable D2H 80 MiB: %.2f ms\n", now() - t); }
    for (int rep = 0; rep < 2; rep++) {
        std::vector<void*> v(16);
        double t = now(); for (auto& q : v) hipMalloc(&q, 40u << 20); double t1 = now();
        for (auto& q : v) hipFree(q); double t2 = now();
        printf("16 x hipMalloc(40 MiB) %.2f ms, 16 x hipFree %.2f ms\n", t1 - t, t2 - t1);
    }
    for (int rep = 0; rep < 2; rep++) {
        std::vector<void*> v(16);
        double t = now(); for (auto& q : v) hipMallocAsync(&q, 40u << 20, s); hipStreamSynchronize(s); double t1 = now();
        for (auto& q : v) hipFreeAsync(q, s); hipStreamSynchronize(s); double t2 = now();
        printf("16 x hipMallocAsync(40 MiB) %.2f ms, 16 x hipFreeAsync %.2f ms\n", t1 - t, t2 - t1);
    }
    { double t = now(); hipStream_t q; hipStreamCreateWithFlags(&q, hipStreamNonBlocking); double t1 = now(); hipStreamDestroy(q); printf("stream create %.2f ms destroy %.2f ms\n", t1 - t, now() - t1); }
    return 0;
}
