// devmem.hpp -- device memory and host <-> device copies of a session.
//
// Plain hipMalloc / hipFree and synchronous copies, deliberately.  A stream-ordered variant (hipMallocAsync / hipFreeAsync on
// the session stream, hipMemcpyAsync from the caller's pageable arrays) took run_poismf's set-up on config C2 from 14 to
// 8 ms -- and lost data on MI355X / ROCm 7.2: in the second and later sessions of a process, uploads into blocks the pool
// had recycled came back all zeros (row pointers, a whole factor) although the stream had been synchronised, and not in
// every run (scripts/probes/probe_det.py shows it as B == 0 after a CG half-sweep).  The set-up is dominated by stream
// creation anyway (7.5 ms per stream, scripts/probes/h2d_probe.hip), which the session avoids by recycling its streams.
// POISMF_HIP_ASYNC_ALLOC=1 switches the stream-ordered allocator back on (development only).
// Round 3: large arrays (16 MB and up) are the exception to "synchronous copies" -- they travel through PINNED chunks with
// hipMemcpyAsync in stream order (second half of this file); the memory itself stays plain hipMalloc, and what misbehaved
// above was the pool allocator together with asynchronous copies from PAGEABLE memory, neither of which this path uses
// (tests/test_gpu_upload.py: same bits with and without it, several sessions per process).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

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

// ---- host -> device through pinned staging ---------------------------------------------------------------------------------
// run_poismf's inputs are the caller's pageable arrays, 2.4 GB of them on the north-star matrix (1.6 GB are size_t indices the
// device keeps as u32).  hipMemcpy from pageable memory moves them at ~25 GB/s through the runtime's own bounce buffers, one
// thread doing the copying; here a few host threads fill pinned chunks -- copying values, or NARROWING indices on the way, so that
// half of the index bytes never cross PCIe -- while the DMA engine drains the chunks filled before (hipMemcpyAsync from pinned
// memory, ~50 GB/s).  Each thread owns a contiguous part of the array and two chunks; an event per chunk says when it may be
// refilled.  One pool PER DEVICE (round 4: the per-device threads of a multi-GPU run_poismf each stage through their own; with one
// pool per process seven of eight set-ups fell back to the pageable path); one staged copy at a time per device, a caller that
// finds its device's pool busy takes the plain path.
struct PmfPinPool {
    static constexpr int THREADS_MAX = 16;
    static constexpr size_t CHUNK = (size_t)4 << 20;
    std::mutex busy;
    void* buf[2 * THREADS_MAX] = {};
    hipEvent_t ev[2 * THREADS_MAX] = {};
    bool ready = false, failed = false;
    // (called with `busy` held and the device current: the events belong to that device)
    bool prepare()
    {
        if (ready || failed) return ready;
        for (int i = 0; i < 2 * THREADS_MAX; i++) {
            if (hipHostMalloc(&buf[i], CHUNK, hipHostMallocDefault) != hipSuccess) buf[i] = nullptr;
            if (buf[i] == nullptr || hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
                // nothing half-built stays behind: this device copies through the plain path from now on
                if (buf[i] != nullptr) { (void)hipHostFree(buf[i]); buf[i] = nullptr; }
                for (int j = 0; j < i; j++) { (void)hipHostFree(buf[j]); buf[j] = nullptr; (void)hipEventDestroy(ev[j]); }
                (void)hipGetLastError();
                failed = true;
                return false;
            }
        }
        ready = true;
        return true;
    }
};
constexpr int PMF_PIN_POOL_DEVICES = 64;
inline PmfPinPool* pmf_pin_pool(int device)
{
    static PmfPinPool pools[PMF_PIN_POOL_DEVICES];
    return device >= 0 && device < PMF_PIN_POOL_DEVICES ? &pools[device] : nullptr;
}
inline int pmf_host_threads()
{
    static const int n = [] {
        int v = 8;
        if (const char* e = getenv("POISMF_HIP_HOST_THREADS")) v = atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && v > hw) v = hw;
        return v < 1 ? 1 : (v > PmfPinPool::THREADS_MAX ? PmfPinPool::THREADS_MAX : v);
    }();
    return n;
}
// arrays below 16 MB are not worth the threads (testing knobs: POISMF_HIP_NO_STAGED_UPLOAD, POISMF_HIP_STAGED_MIN_BYTES)
inline bool pmf_staged_wanted(size_t bytes)
{
    static const bool off = getenv("POISMF_HIP_NO_STAGED_UPLOAD") != nullptr;
    static const size_t least = getenv("POISMF_HIP_STAGED_MIN_BYTES") ? (size_t)atoll(getenv("POISMF_HIP_STAGED_MIN_BYTES")) : ((size_t)16 << 20);
    return !off && bytes >= least && bytes > 0;
}
// dst[0 .. n) items of `item` bytes each; fill(pinned, first_item, count) writes count items.  Returns hipErrorNotReady when the
// staged path is not available (pool busy / no pinned memory / small array): the caller then uses pmf_upload.
template <class Fill> inline hipError_t pmf_upload_staged(void* dst, size_t n, size_t item, int device, hipStream_t stream, Fill&& fill)
{
    if (!pmf_staged_wanted(n * item)) return hipErrorNotReady;
    PmfPinPool* poolp = pmf_pin_pool(device);
    if (poolp == nullptr) return hipErrorNotReady;
    PmfPinPool& pool = *poolp;
    std::unique_lock<std::mutex> lk(pool.busy, std::try_to_lock);
    if (!lk.owns_lock() || hipSetDevice(device) != hipSuccess || !pool.prepare()) return hipErrorNotReady;
    const int nt = pmf_host_threads();
    const size_t per_chunk = PmfPinPool::CHUNK / item;
    std::vector<hipError_t> err((size_t)nt, hipSuccess);
    auto work = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { err[(size_t)t] = hipErrorInvalidDevice; return; }
        const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
        int b = 0;
        for (size_t i = lo; i < hi; i += per_chunk, b ^= 1) {
            const size_t cnt = hi - i < per_chunk ? hi - i : per_chunk;
            const int slot = 2 * t + b;
            hipError_t e = hipEventSynchronize(pool.ev[slot]);   // (a never-recorded event is complete)
            if (e == hipSuccess) {
                fill(pool.buf[slot], i, cnt);
                e = hipMemcpyAsync((char*)dst + i * item, pool.buf[slot], cnt * item, hipMemcpyHostToDevice, stream);
            }
            if (e == hipSuccess) e = hipEventRecord(pool.ev[slot], stream);
            if (e != hipSuccess) { err[(size_t)t] = e; return; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (hipError_t e : err) if (e != hipSuccess) return e;
    return hipSuccess;   // (in stream order; the chunks stay owned by the pool until their events complete)
}

// The way back (the factors after the last sweep): chunks land in pinned memory and the threads copy them out to the caller's
// array while the next chunk of each is in flight.  take(pinned, first_item, count) consumes count items.
template <class Take> inline hipError_t pmf_download_staged(const void* src, size_t n, size_t item, int device, hipStream_t stream, Take&& take)
{
    if (!pmf_staged_wanted(n * item)) return hipErrorNotReady;
    PmfPinPool* poolp = pmf_pin_pool(device);
    if (poolp == nullptr) return hipErrorNotReady;
    PmfPinPool& pool = *poolp;
    std::unique_lock<std::mutex> lk(pool.busy, std::try_to_lock);
    if (!lk.owns_lock() || hipSetDevice(device) != hipSuccess || !pool.prepare()) return hipErrorNotReady;
    const int nt = pmf_host_threads();
    const size_t per_chunk = PmfPinPool::CHUNK / item;
    std::vector<hipError_t> err((size_t)nt, hipSuccess);
    auto work = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { err[(size_t)t] = hipErrorInvalidDevice; return; }
        const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
        auto issue = [&](size_t i, int b) -> hipError_t {
            const size_t cnt = hi - i < per_chunk ? hi - i : per_chunk;
            hipError_t e = hipMemcpyAsync(pool.buf[2 * t + b], (const char*)src + i * item, cnt * item, hipMemcpyDeviceToHost, stream);
            return e != hipSuccess ? e : hipEventRecord(pool.ev[2 * t + b], stream);
        };
        hipError_t e = hipEventSynchronize(pool.ev[2 * t]);   // (an upload's chunk may still be draining)
        if (e == hipSuccess) e = hipEventSynchronize(pool.ev[2 * t + 1]);
        if (e == hipSuccess && lo < hi) e = issue(lo, 0);
        int b = 0;
        for (size_t i = lo; i < hi && e == hipSuccess; i += per_chunk, b ^= 1) {
            if (i + per_chunk < hi) e = issue(i + per_chunk, b ^ 1);
            if (e == hipSuccess) e = hipEventSynchronize(pool.ev[2 * t + b]);
            if (e == hipSuccess) take(pool.buf[2 * t + b], i, hi - i < per_chunk ? hi - i : per_chunk);
        }
        err[(size_t)t] = e;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (hipError_t e : err) if (e != hipSuccess) return e;
    return hipSuccess;   // (complete: every chunk was waited for)
}
// A plain array either way, staged when it can be.
inline hipError_t pmf_upload_big(void* dst, const void* src, size_t bytes, int device, hipStream_t stream)
{
    const char* from = (const char*)src;
    const hipError_t e = pmf_upload_staged(dst, bytes, 1, device, stream, [from](void* pin, size_t i0, size_t cnt) { memcpy(pin, from + i0, cnt); });
    return e == hipErrorNotReady ? pmf_upload(dst, src, bytes, stream) : e;
}
inline hipError_t pmf_download_big(void* dst, const void* src, size_t bytes, int device, hipStream_t stream)
{
    char* to = (char*)dst;
    const hipError_t e = pmf_download_staged(src, bytes, 1, device, stream, [to](void* pin, size_t i0, size_t cnt) { memcpy(to + i0, pin, cnt); });
    return e == hipErrorNotReady ? pmf_download(dst, src, bytes, stream) : e;
}
