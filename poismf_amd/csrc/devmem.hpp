// devmem.hpp -- device memory of a session is stream-ordered (hipMallocAsync / hipFreeAsync on the session stream):
// measured on MI355X / ROCm 7.2, hipFree costs 170 us per buffer (it synchronises the device), hipFreeAsync 30 us --
// with ~30 buffers per run_poismf call that was 6 ms of a 25 ms call on config C2 (scripts/probes/h2d_probe.hip).
#pragma once
#include <hip/hip_runtime.h>

template <class T> inline hipError_t pmf_alloc(T** p, size_t bytes, hipStream_t stream)
{
    void* q = nullptr;
    const hipError_t e = hipMallocAsync(&q, bytes ? bytes : 16, stream);
    *p = (T*)q;
    return e;
}
inline void pmf_free(void* p, hipStream_t stream)
{
    if (p != nullptr) (void)hipFreeAsync(p, stream);
}
