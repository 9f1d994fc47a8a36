// serve.hip -- the two serving-side helpers of the reference:
//
//   predict_multiple   out[i] = A[ixA[i], :] . B[ixB[i], :]                 ref: src/pred.c:42-64
//   topN               indices (and scores) of the n_top largest a . B[j, :] over an include list, or over all
//                      items minus an exclude list, sorted by score descending  ref: src/topN.c:112-284
//
// Both are bandwidth-bound gathers / streams; the ranking is a descending radix sort of (score, index) pairs
// (rocPRIM), which also fixes the order among equal scores (ascending index) that the reference leaves to qsort.
// Exported under the reference's names with the reference's signatures (include/poismf_hip.h).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "../../include/poismf_hip.h"
#include "devmem.hpp"

namespace {

// one thread per (row of P, row of Q) pair; k-ordered FMA chain like the reference's ddot
__global__ void pair_dot_kernel(const real_t* P, const real_t* Q, const unsigned* ixP, const unsigned* ixQ, size_t n, int k,
                                real_t* out)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const real_t* p = P + (size_t)ixP[i] * (size_t)k;
        const real_t* q = Q + (size_t)ixQ[i] * (size_t)k;
        real_t s = 0;
        for (int c = 0; c < k; c++) s += p[c] * q[c];
        out[i] = s;
    }
}

// scores[i] = a . B[cand[i], :]   (cand == nullptr: i itself); one 16-lane group per item, coalesced row reads
__global__ void score_kernel(const real_t* a, const real_t* B, const unsigned* cand, size_t n, int k, real_t* scores,
                             unsigned* ids)
{
    const int sub = threadIdx.x & 15;
    const size_t grp = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4;
    const size_t ngrp = ((size_t)gridDim.x * blockDim.x) >> 4;
    for (size_t i = grp; i < n; i += ngrp) {
        const unsigned j = cand ? cand[i] : (unsigned)i;
        const real_t* row = B + (size_t)j * (size_t)k;
        real_t s = 0;
        for (int c = sub; c < k; c += 16) s += a[c] * row[c];
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
        if (sub == 0) { scores[i] = s; ids[i] = j; }
    }
}
__global__ void mask_kernel(const unsigned* excl, size_t n_excl, real_t* scores)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_excl; i += (size_t)gridDim.x * blockDim.x)
        scores[excl[i]] = -std::numeric_limits<real_t>::infinity();
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return pmf_malloc_retry(&p, bytes ? bytes : 16) == hipSuccess; }
    template <class U> U* as() { return (U*)p; }
};

int pick_device()
{
    int device = 0;
    if (const char* e = getenv("POISMF_HIP_DEVICE")) device = atoi(e);
    return device;
}

bool upload_u32(DevBuf& d, const sparse_ix* h, size_t n, size_t* maxv)
{
    std::vector<unsigned> t(n ? n : 1);
    size_t m = 0;
    for (size_t i = 0; i < n; i++) { t[i] = (unsigned)h[i]; m = std::max(m, (size_t)h[i]); }
    if (maxv) *maxv = m;
    return d.alloc(sizeof(unsigned) * n) && hipMemcpy(d.p, t.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice) == hipSuccess;
}

}  // namespace

// ---- cores on device-resident factors (the drop-ins below copy the caller's factors up first; a session already has them) ----
// out[i] = A[ixA[i]] . B[ixB[i]] for host index arrays; A / B are device pointers.  Returns 0 / 1.
int poismf_hip_serve_predict(const real_t* dA, const real_t* dB, const sparse_ix* ixA, const sparse_ix* ixB, size_t n, int k, real_t* out,
                             size_t* max_a, size_t* max_b)
{
    DevBuf dia, dib, dout;
    if (!upload_u32(dia, ixA, n, max_a) || !upload_u32(dib, ixB, n, max_b) || !dout.alloc(n * sizeof(real_t))) return 1;
    if (dA == nullptr || dB == nullptr) return 0;   // (index upload only: the drop-in sizes its factor copies from the maxima)
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(pair_dot_kernel, dim3(grid), dim3(256), 0, nullptr, dA, dB, dia.as<unsigned>(), dib.as<unsigned>(), n, k,
                       dout.as<real_t>());
    return hipMemcpy(out, dout.p, n * sizeof(real_t), hipMemcpyDeviceToHost) != hipSuccess;
}

// Top-N of a . B[j] over the candidates; d_a (k reals) and dB ([n x k]) are device pointers; index lists are host arrays.
int poismf_hip_serve_topn(const real_t* d_a, const real_t* dB, int k, const sparse_ix* include_ix, size_t n_include,
                          const sparse_ix* exclude_ix, size_t n_exclude, sparse_ix* outp_ix, real_t* outp_score, size_t n_top, size_t n)
{
    const size_t n_cand = include_ix ? n_include : n;
    DevBuf dcand, dexcl, dsc, dsc2, did, did2, dtmp;
    if (include_ix && !upload_u32(dcand, include_ix, n_include, nullptr)) return 1;
    if (exclude_ix && !upload_u32(dexcl, exclude_ix, n_exclude, nullptr)) return 1;
    if (!dsc.alloc(sizeof(real_t) * n_cand) || !dsc2.alloc(sizeof(real_t) * n_cand) || !did.alloc(sizeof(unsigned) * n_cand) ||
        !did2.alloc(sizeof(unsigned) * n_cand))
        return 1;
    const unsigned grid = (unsigned)std::min<size_t>((n_cand * 16 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(score_kernel, dim3(grid), dim3(256), 0, nullptr, d_a, dB, include_ix ? dcand.as<unsigned>() : (const unsigned*)nullptr,
                       n_cand, k, dsc.as<real_t>(), did.as<unsigned>());
    if (exclude_ix) {
        const unsigned g2 = (unsigned)std::min<size_t>((n_exclude + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(mask_kernel, dim3(g2), dim3(256), 0, nullptr, dexcl.as<unsigned>(), n_exclude, dsc.as<real_t>());
    }
    // stable descending sort: equal scores keep ascending candidate order
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, dsc.as<real_t>(), dsc2.as<real_t>(), did.as<unsigned>(), did2.as<unsigned>(),
                                       n_cand, 0u, (unsigned)(8 * sizeof(real_t)), (hipStream_t) nullptr) != hipSuccess)
        return 1;
    if (!dtmp.alloc(tmp_bytes)) return 1;
    if (rocprim::radix_sort_pairs_desc(dtmp.p, tmp_bytes, dsc.as<real_t>(), dsc2.as<real_t>(), did.as<unsigned>(), did2.as<unsigned>(),
                                       n_cand, 0u, (unsigned)(8 * sizeof(real_t)), (hipStream_t) nullptr) != hipSuccess)
        return 1;
    std::vector<unsigned> hid(n_top);
    if (hipMemcpy(hid.data(), did2.p, sizeof(unsigned) * n_top, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (size_t i = 0; i < n_top; i++) outp_ix[i] = (sparse_ix)hid[i];
    if (outp_score != nullptr &&
        hipMemcpy(outp_score, dsc2.p, sizeof(real_t) * n_top, hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    return 0;
}

// the argument checks of ref src/topN.c:126-130 (2 = invalid combination)
int poismf_hip_serve_topn_check(const sparse_ix*& include_ix, size_t n_include, const sparse_ix*& exclude_ix, size_t n_exclude, size_t n_top, size_t n)
{
    if (n_include == 0) include_ix = nullptr;
    if (n_exclude == 0) exclude_ix = nullptr;
    if (include_ix != nullptr && exclude_ix != nullptr) return 2;
    if (n_top == 0) return 2;
    if (n_exclude > n - n_top) return 2;
    if (n_include > n) return 2;
    if (n_top > (include_ix ? n_include : n)) return 2;
    return 0;
}

extern "C" {

void predict_multiple(real_t* out, real_t* A, real_t* B, sparse_ix* ixA, sparse_ix* ixB, size_t n, int k, int nthreads)
{
    (void)nthreads;
    if (n == 0) return;
    // the reference returns void: on failure the outputs are filled with NaN and a message goes to stderr
    auto fail = [&]() {
        fprintf(stderr, "Error: out of memory.\n");
        for (size_t i = 0; i < n; i++) out[i] = std::numeric_limits<real_t>::quiet_NaN();
    };
    if (hipSetDevice(pick_device()) != hipSuccess) return fail();
    size_t ma = 0, mb = 0;
    for (size_t i = 0; i < n; i++) { ma = std::max(ma, (size_t)ixA[i]); mb = std::max(mb, (size_t)ixB[i]); }
    DevBuf dA, dB;
    const size_t ba = (ma + 1) * (size_t)k * sizeof(real_t), bb = (mb + 1) * (size_t)k * sizeof(real_t);
    if (!dA.alloc(ba) || !dB.alloc(bb) ||
        hipMemcpy(dA.p, A, ba, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dB.p, B, bb, hipMemcpyHostToDevice) != hipSuccess ||
        poismf_hip_serve_predict(dA.as<real_t>(), dB.as<real_t>(), ixA, ixB, n, k, out, nullptr, nullptr))
        return fail();
}

int topN(real_t* a_vec, real_t* B, int k, sparse_ix* include_ix, size_t n_include, sparse_ix* exclude_ix, size_t n_exclude,
         sparse_ix* outp_ix, real_t* outp_score, size_t n_top, size_t n, int nthreads)
{
    (void)nthreads;
    const sparse_ix *inc = include_ix, *exc = exclude_ix;
    if (const int rc = poismf_hip_serve_topn_check(inc, n_include, exc, n_exclude, n_top, n)) return rc;   // ref: :126-130
    if (hipSetDevice(pick_device()) != hipSuccess) return 1;
    size_t maxrow = n - 1;
    if (inc) { maxrow = 0; for (size_t i = 0; i < n_include; i++) maxrow = std::max(maxrow, (size_t)inc[i]); }
    const size_t nrowsB = inc ? maxrow + 1 : n;
    DevBuf da, dB;
    if (!da.alloc(sizeof(real_t) * k) || !dB.alloc(sizeof(real_t) * nrowsB * k) ||
        hipMemcpy(da.p, a_vec, sizeof(real_t) * k, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dB.p, B, sizeof(real_t) * nrowsB * k, hipMemcpyHostToDevice) != hipSuccess)
        return 1;
    return poismf_hip_serve_topn(da.as<real_t>(), dB.as<real_t>(), k, inc, n_include, exc, n_exclude, outp_ix, outp_score, n_top, n);
}

}  // extern "C"
