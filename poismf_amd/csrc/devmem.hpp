// devmem.hpp -- device memory and host <-> device copies of a session.
//
// Plain hipMalloc / hipFree and synchronous copies, deliberately.  A stream-ordered variant (hipMallocAsync / hipFreeAsync on
// the session stream, hipMemcpyAsync from the caller's pageable arrays) took run_poismf's set-up on config C2 from 14 to
// 8 ms -- and lost data on MI355X / ROCm 7.2: in the second and later sessions of a process, uploads into blocks the pool
// had recycled came back all zeros (row pointers, a whole factor) although the stream had been synchronised, and not in
// every run (scripts/probes/probe_det.py shows it as B == 0 after a CG half-sweep).  The set-up is dominated by stream
// creation anyway (7.5 ms per stream, scripts/probes/h2d_probe.hip), which the session avoids by recycling its streams.
// (The switch that brought the stream-ordered allocator back, POISMF_HIP_ASYNC_ALLOC, went in round 6 with the code behind it.)
// Round 3: large arrays (16 MB and up) are the exception to "synchronous copies" -- they travel through PINNED chunks with
// hipMemcpyAsync in stream order (second half of this file); the memory itself stays plain hipMalloc, and what misbehaved
// above was the pool allocator together with asynchronous copies from PAGEABLE memory, neither of which this path uses
// (tests/test_gpu_upload.py: same bits with and without it, several sessions per process).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

// POISMF_HIP_VERBOSE=2: host wall-clock stamps of the set-up / copy phases on stderr as well (development aid; the stamps are taken where
// the host is, asynchronous work may still be in flight behind them)
inline void pmf_tl(const char* what)
{
    static const bool on = getenv("POISMF_HIP_VERBOSE") != nullptr && atoi(getenv("POISMF_HIP_VERBOSE")) >= 2;
    if (!on) return;
    static double first = 0, last = 0;
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const double t = (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
    if (what == nullptr || first == 0) first = last = t;
    if (what != nullptr) fprintf(stderr, "[tl %8.2f +%7.2f] %s\n", t - first, t - last, what);
    last = t;
}

// ---- the device arrays of finished sessions are kept for the next one (round 4) ---------------------------------------------------
// run_poismf is a one-shot call: 2.5 GB of device arrays allocated, filled, used and freed per fit.  On this stack the freeing and
// allocating is not only 2.5 ms of calls: it also leaves the copy engines busy behind the call's back (scripts/probes/stage_probe2.hip --
// staged uploads over two DMA queues move 400 MB in 7.2-7.9 ms into arrays that stay allocated and in 11.5-12 ms, slower than one
// queue, while arrays of that size are being allocated and freed around them).  So blocks of 1 MB and more go to a per-process list
// when they are released and come back to the next request of exactly their size on their device: repeated fits on matrices of one
// shape -- a hyper-parameter search, the bench's run_poismf leg -- find all their arrays there.  The list holds at most
// POISMF_HIP_DEVICE_CACHE_MB / poismf_hip_set_device_cache_mb(); oldest blocks leave first; poismf_hip_release_cache() empties it;
// an allocation that fails empties it and tries again.  Blocks come back with their old contents (as hipMalloc's are unspecified):
// nothing here may rely on fresh memory reading as zero.
// OFF BY DEFAULT since round 5 (limit 0: plain hipMalloc / hipFree): the reference frees everything before run_poismf returns (ref:
// src/poismf.c:610-617, SURVEY 8b "no handles/state survive the call") and so does this library unless the caller opts in -- round 4
// kept up to 16 GB per process (per flavour library) alive behind the caller's back.  The limit can be changed at run time.
struct PmfDevCache {
    struct Block { void* p; size_t bytes; int device; };
    std::mutex mu;
    std::vector<Block> idle;                                        // released, oldest first
    std::unordered_map<void*, std::pair<size_t, int>> live;         // handed out, eligible to come back
    size_t idle_bytes = 0;
    static constexpr size_t LEAST = (size_t)1 << 20;
    static std::atomic<size_t>& limit_word()
    {
        static std::atomic<size_t> v{ [] {
            const char* e = getenv("POISMF_HIP_DEVICE_CACHE_MB");
            return (size_t)(e ? atoll(e) : 0) << 20;
        }() };
        return v;
    }
    static size_t limit() { return limit_word().load(std::memory_order_relaxed); }
    // (mu held) takes the oldest blocks off the list until at most `room` bytes are left; the caller hands them to hipFree AFTER
    // it has released the mutex (hipFree waits for the device: other threads' allocations must not queue behind it)
    std::vector<void*> take_oldest_until(size_t room)
    {
        std::vector<void*> out;
        size_t n = 0;
        while (n < idle.size() && idle_bytes > room) { out.push_back(idle[n].p); idle_bytes -= idle[n].bytes; n++; }
        idle.erase(idle.begin(), idle.begin() + (long)n);
        return out;
    }
};
inline PmfDevCache& pmf_dev_cache()
{
    static PmfDevCache* c = new PmfDevCache();   // (never destroyed: no HIP calls from static destructors at exit)
    return *c;
}
inline void pmf_release_cache()
{
    PmfDevCache& c = pmf_dev_cache();
    std::vector<void*> gone;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        gone = c.take_oldest_until(0);
    }
    for (void* p : gone) (void)hipFree(p);
}
// new limit in MB (0 = keep nothing: what is on the list now is freed); returns the old one
inline size_t pmf_set_cache_limit_mb(size_t mb)
{
    const size_t old = PmfDevCache::limit_word().exchange(mb << 20, std::memory_order_relaxed) >> 20;
    PmfDevCache& c = pmf_dev_cache();
    std::vector<void*> gone;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        gone = c.take_oldest_until(mb << 20);
    }
    for (void* p : gone) (void)hipFree(p);
    return old;
}
// hipMalloc for the places that do not go through pmf_alloc (temporaries of the COO conversion, the serving calls): an out-of-memory
// answer gives the kept blocks back to the driver and asks once more
inline hipError_t pmf_malloc_retry(void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        pmf_release_cache();
        e = hipMalloc(p, bytes);
    }
    return e;
}

template <class T> inline hipError_t pmf_alloc(T** p, size_t bytes, hipStream_t stream)
{
    void* q = nullptr;
    if (bytes == 0) bytes = 16;
    PmfDevCache& c = pmf_dev_cache();
    const bool cached = bytes >= PmfDevCache::LEAST && PmfDevCache::limit() > 0;
    int device = 0;
    if (cached) {
        if (hipGetDevice(&device) != hipSuccess) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lk(c.mu);
        for (size_t i = c.idle.size(); i-- > 0;) {
            if (c.idle[i].bytes == bytes && c.idle[i].device == device) {
                q = c.idle[i].p;
                c.idle_bytes -= bytes;
                c.idle.erase(c.idle.begin() + (long)i);
                c.live[q] = std::make_pair(bytes, device);
                *p = (T*)q;
                return hipSuccess;
            }
        }
    }
    hipError_t e = hipMalloc(&q, bytes);
    if (e == hipErrorOutOfMemory) {   // make room: whatever the cache holds goes back to the driver first
        (void)hipGetLastError();
        pmf_release_cache();
        q = nullptr;
        e = hipMalloc(&q, bytes);
    }
    if (e == hipSuccess && cached) {
        std::lock_guard<std::mutex> lk(c.mu);
        c.live[q] = std::make_pair(bytes, device);
    }
    *p = (T*)q;
    return e;
}
inline void pmf_free(void* p, hipStream_t stream)
{
    if (p == nullptr) return;
    PmfDevCache& c = pmf_dev_cache();
    {
        std::unique_lock<std::mutex> lk(c.mu);
        const auto it = c.live.find(p);
        if (it != c.live.end()) {
            const size_t bytes = it->second.first;
            const int device = it->second.second;
            c.live.erase(it);
            if (bytes <= PmfDevCache::limit()) {
                lk.unlock();
                // what hipFree does implicitly and the callers rely on: nothing on the device can still be using the block
                int cur = -1;
                (void)hipGetDevice(&cur);
                if (cur != device) (void)hipSetDevice(device);
                (void)hipDeviceSynchronize();
                if (cur != device && cur >= 0) (void)hipSetDevice(cur);
                lk.lock();
                const size_t lim = PmfDevCache::limit();   // (may have been lowered meanwhile)
                if (bytes <= lim) {
                    std::vector<void*> gone = c.take_oldest_until(lim - bytes);
                    c.idle.push_back({ p, bytes, device });
                    c.idle_bytes += bytes;
                    lk.unlock();
                    for (void* q : gone) (void)hipFree(q);
                    return;
                }
                lk.unlock();
                (void)hipFree(p);
                return;
            }
        }
    }
    (void)hipFree(p);   // waits for the device: nothing can still be using the block
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
// refilled.  Round 4: TWO DMA queues -- the caller's stream and the pool's own side stream, threads alternating between them, the
// side stream fenced against the caller's by an event at either end: one queue moves 4 MB chunks at 41-44 GB/s, two at 53-55
// (scripts/probes/stage_probe.hip: 400 MB in 10.1 -> 7.9 ms with the narrowing fill; 16 MB chunks, 16 threads or four queues add nothing).
// One pool PER DEVICE (round 4: the per-device threads of a multi-GPU run_poismf each stage through their own; with one
// pool per process seven of eight set-ups fell back to the pageable path); one staged copy at a time per device, a caller that
// finds its device's pool busy takes the plain path.
struct PmfPinPool {
    static constexpr int THREADS_MAX = 16;
    static constexpr size_t CHUNK = (size_t)4 << 20;
    std::mutex busy;
    void* buf[2 * THREADS_MAX] = {};
    hipEvent_t ev[2 * THREADS_MAX] = {};
    hipStream_t side = nullptr;          // the second DMA queue
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool ready = false, failed = false;
    // (called with `busy` held and the device current: the events and the side stream belong to that device)
    bool prepare()
    {
        if (ready || failed) return ready;
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_begin, hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&ev_end, hipEventDisableTiming) != hipSuccess) {
            if (ev_begin != nullptr) (void)hipEventDestroy(ev_begin);
            if (side != nullptr) (void)hipStreamDestroy(side);
            side = nullptr; ev_begin = ev_end = nullptr;
            (void)hipGetLastError();
            failed = true;
            return false;
        }
        for (int i = 0; i < 2 * THREADS_MAX; i++) {
            if (hipHostMalloc(&buf[i], CHUNK, hipHostMallocDefault) != hipSuccess) buf[i] = nullptr;
            if (buf[i] == nullptr || hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
                // nothing half-built stays behind: this device copies through the plain path from now on
                if (buf[i] != nullptr) { (void)hipHostFree(buf[i]); buf[i] = nullptr; }
                for (int j = 0; j < i; j++) { (void)hipHostFree(buf[j]); buf[j] = nullptr; (void)hipEventDestroy(ev[j]); }
                (void)hipEventDestroy(ev_begin); (void)hipEventDestroy(ev_end); (void)hipStreamDestroy(side);
                side = nullptr; ev_begin = ev_end = nullptr;
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
// the second DMA queue always joins in (one queue alone -- POISMF_HIP_ONE_DMA_QUEUE, rounds 4-5 -- moved 400 MB in 11.5 instead of 7.5 ms)
inline bool pmf_two_queues() { return true; }
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
    // odd threads copy on the side queue, which starts behind everything `stream` holds now and which `stream` waits for at the end
    const bool two = pmf_two_queues() && nt > 1;
    if (two) {
        hipError_t e = hipEventRecord(pool.ev_begin, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(pool.side, pool.ev_begin, 0);
        if (e != hipSuccess) return e;
    }
    auto work = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { err[(size_t)t] = hipErrorInvalidDevice; return; }
        const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
        const hipStream_t q = (two && (t & 1)) ? pool.side : stream;
        int b = 0;
        for (size_t i = lo; i < hi; i += per_chunk, b ^= 1) {
            const size_t cnt = hi - i < per_chunk ? hi - i : per_chunk;
            const int slot = 2 * t + b;
            hipError_t e = hipEventSynchronize(pool.ev[slot]);   // (a never-recorded event is complete)
            if (e == hipSuccess) {
                fill(pool.buf[slot], i, cnt);
                e = hipMemcpyAsync((char*)dst + i * item, pool.buf[slot], cnt * item, hipMemcpyHostToDevice, q);
            }
            if (e == hipSuccess) e = hipEventRecord(pool.ev[slot], q);
            if (e != hipSuccess) { err[(size_t)t] = e; return; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    if (two) {   // (also after a failure: `stream` must not run ahead of copies still queued on the side)
        hipError_t e = hipEventRecord(pool.ev_end, pool.side);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, pool.ev_end, 0);
        if (e != hipSuccess) return e;
    }
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
    const bool two = pmf_two_queues() && nt > 1;
    if (two) {   // the side queue starts behind whatever on `stream` produces the data
        hipError_t e = hipEventRecord(pool.ev_begin, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(pool.side, pool.ev_begin, 0);
        if (e != hipSuccess) return e;
    }
    auto work = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { err[(size_t)t] = hipErrorInvalidDevice; return; }
        const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
        const hipStream_t q = (two && (t & 1)) ? pool.side : stream;
        auto issue = [&](size_t i, int b) -> hipError_t {
            const size_t cnt = hi - i < per_chunk ? hi - i : per_chunk;
            hipError_t e = hipMemcpyAsync(pool.buf[2 * t + b], (const char*)src + i * item, cnt * item, hipMemcpyDeviceToHost, q);
            return e != hipSuccess ? e : hipEventRecord(pool.ev[2 * t + b], q);
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
    for (hipError_t e : err) if (e != hipSuccess) { if (two) (void)hipStreamSynchronize(pool.side); return e; }
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
