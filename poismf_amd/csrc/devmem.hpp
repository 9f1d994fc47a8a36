// devmem.hpp -- device memory and host <-> device copies of a session.
//
// Plain hipMalloc / hipFree and synchronous copies, deliberately.  A stream-ordered variant (hipMallocAsync / hipFreeAsync on
// the session stream, hipMemcpyAsync from the caller's pageable arrays) took run_poismf's set-up on config C2 from 14 to
// 8 ms -- and lost data on MI355X / ROCm 7.2: in the second and later sessions of a process, uploads into blocks the pool
// had recycled came back all zeros (row pointers, a whole factor) although the stream had been synchronised, and not in
// every run (scripts/probes/probe_det.py shows it as B == 0 after a CG half-sweep).  The set-up is dominated by stream
// creation anyway (7.5 ms per stream, scripts/probes/h2d_probe.hip), which the session avoids by recycling its streams.
// POISMF_HIP_ASYNC_ALLOC=1 switches the stream-ordered allocator back on (development only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

inline bool pmf_async_alloc()
{
    static const bool on = getenv("POISMF_HIP_ASYNC_ALLOC") != nullptr;
    return on;
}

template <class T> inline hipError_t pmf_alloc(T** p, size_t bytes, hipStream_t stream)
{
    void* q = nullptr;
    const hipError_t e = pmf_async_alloc() ? hipMallocAsync(&q, bytes ? bytes : 16, stream) : hipMalloc(&q, bytes ? bytes : 16);
    *p = (T*)q;
    return e;
}
inline void pmf_free(void* p, hipStream_t stream)
{
    if (p == nullptr) return;
    if (pmf_async_alloc()) (void)hipFreeAsync(p, stream);
    else (void)hipFree(p);   // waits for the device: nothing can still be using the block
}

// Host <-> device copies of CALLER-OWNED (pageable) memory: drain the stream, then copy synchronously; kernels launched
// afterwards on any stream see the data.
inline hipError_t pmf_upload(void* dst, const void* src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    const hipError_t e = hipStreamSynchronize(stream);
    return e != hipSuccess ? e : hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}
inline hipError_t pmf_download(void* dst, const void* src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    const hipError_t e = hipStreamSynchronize(stream);
    return e != hipSuccess ? e : hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
}
