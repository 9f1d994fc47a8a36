// waitvalue_probe.hip -- does hipStreamWaitValue32 on plain device memory let stream B start a kernel as soon as kernel A (still
// running on stream A) has written a counter?   hipcc --offload-arch=gfx950 -O3 scripts/probes/waitvalue_probe.hip -o /tmp/wv && /tmp/wv
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void a_kernel(unsigned* counter, unsigned long long* stamp, long long spin)
{
    if (threadIdx.x == 0) { atomicAdd(counter, 1u); }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = wall_clock64();   // when A ended
}
__global__ void b_kernel(unsigned long long* stamp) { if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64(); }   // when B ran
int main()
{
    unsigned* counter; unsigned long long* stamp;
    hipMalloc(&counter, 4); hipMalloc(&stamp, 16); hipMemset(counter, 0, 4); hipMemset(stamp, 0, 16);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipLaunchKernelGGL(a_kernel, dim3(60), dim3(512), 0, sa, counter, stamp, 100000000ll / 10);   // 100 MHz clock: 0.1 s
    hipError_t e = hipStreamWaitValue32(sb, counter, 60, hipStreamWaitValueGte, 0xffffffffu);
    printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e));
    hipLaunchKernelGGL(b_kernel, dim3(1), dim3(64), 0, sb, stamp);
    e = hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost);
    printf("sync: %s; A ended at %llu, B ran at %llu: B %s A's end (%.3f ms apart)\n", hipGetErrorString(e), h[0], h[1], h[1] < h[0] ? "BEFORE" : "after",
           (double)((long long)h[0] - (long long)h[1]) / 1e5);
    return 0;
}
