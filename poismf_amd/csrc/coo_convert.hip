// coo_convert.hip -- COO -> CSR and CSC on the device, duplicates summed, indices sorted within rows.
//
// SURVEY.md section 8f, N3.  Replaces the host conversion the reference's Python class performs before
// every fit (ref: poismf/__init__.py:404-414: coo.tocsr() and coo.tocsc(), which sum duplicate (i,j)
// entries and deliver sorted indices): at 1e8 triplets SciPy needs several seconds per orientation while
// one GPU sweep takes 0.2 s.  Pure HBM-bound integer work: pack (major, minor) into a 64-bit key, LSD radix
// sort the (key, value) pairs (rocPRIM's device radix sort -- a library primitive, used the way a library
// GEMM would be), sum equal keys, split the keys again and binary-search the row pointers.
//
// Compiled into libpoismf_hip_{d,f}.so next to poismf_hip.hip; C-ABI in include/poismf_hip.h.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/poismf_hip.h"
#include "devmem.hpp"

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "poismf_hip: %s failed: %s\n", #expr, hipGetErrorString(e_));          \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

// Triplets whose major index lies outside [m0, m1) get the all-ones key: they sort behind every real key, collapse
// into one trailing entry in the reduction and are dropped there (row shards of the multi-GPU driver).
__global__ void pack_keys(const unsigned* major, const unsigned* minor, size_t n, unsigned m0, unsigned m1, unsigned long long* keys)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned m = major[i];
        keys[i] = (m >= m0 && m < m1) ? (((unsigned long long)(m - m0) << 32) | (unsigned long long)minor[i]) : ~0ull;
    }
}
__global__ void row_len_kernel(const unsigned long long* indptr, size_t n, unsigned base, unsigned* len, unsigned* ids)
{
    for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < n; r += (size_t)gridDim.x * blockDim.x) {
        len[r] = (unsigned)(indptr[r + 1] - indptr[r]);
        ids[r] = base + (unsigned)r;
    }
}
__global__ void narrow_kernel(const unsigned long long* src, size_t n, unsigned* dst)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (unsigned)src[i];
}
__global__ void split_keys(const unsigned long long* keys, size_t n, unsigned* minor)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        minor[i] = (unsigned)(keys[i] & 0xffffffffull);
}
// indptr[r] = number of unique keys whose major index is < r  (lower bound of r << 32)
__global__ void row_pointers(const unsigned long long* keys, size_t n, size_t dim, unsigned long long* indptr)
{
    for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r <= dim; r += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long target = (unsigned long long)r << 32;
        size_t lo = 0, hi = n;
        while (lo < hi) {
            const size_t mid = lo + (hi - lo) / 2;
            if (keys[mid] < target) lo = mid + 1; else hi = mid;
        }
        indptr[r] = lo;
    }
}

int bits_for(size_t v)
{
    int b = 1;
    while (b < 32 && ((size_t)1 << b) < v) b++;
    return b;
}

}  // namespace

// One orientation, everything on the device.  d_major / d_minor / d_val: n triplets, of which those with major index in
// [major_begin, major_end) are kept (rebased to major_begin).  Outputs (device, capacity n): out_minor (u32), out_val,
// out_indptr (major_end - major_begin + 1, u64); *nnz_out = number of distinct (major, minor) pairs kept.
// Scratch is allocated and freed inside.  Also called by poismf_hip.hip (session creation from COO).
int poismf_hip_device_coo_to_cs(const unsigned* d_major, const unsigned* d_minor, const real_t* d_val, size_t n, size_t major_begin,
                                size_t major_end, unsigned* out_minor, real_t* out_val, unsigned long long* out_indptr,
                                size_t* nnz_out, hipStream_t stream)
{
    const size_t dim_major = major_end - major_begin;
    unsigned long long *keys_a = nullptr, *keys_b = nullptr;
    real_t* vals_b = nullptr;
    size_t* d_count = nullptr;
    void* tmp = nullptr;
    auto cleanup = [&]() {
        pmf_free(keys_a, stream);
        pmf_free(keys_b, stream);
        pmf_free(vals_b, stream);
        pmf_free(d_count, stream);
        pmf_free(tmp, stream);
    };
#define TRY_OR_CLEAN(expr) do { if ((expr) != hipSuccess) { fprintf(stderr, "poismf_hip: %s failed\n", #expr); cleanup(); return 1; } } while (0)
    TRY_OR_CLEAN(pmf_alloc(&keys_a, sizeof(unsigned long long) * n, stream));
    TRY_OR_CLEAN(pmf_alloc(&keys_b, sizeof(unsigned long long) * n, stream));
    TRY_OR_CLEAN(pmf_alloc(&vals_b, sizeof(real_t) * n, stream));
    TRY_OR_CLEAN(pmf_alloc(&d_count, sizeof(size_t), stream));
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(pack_keys, dim3(grid), dim3(256), 0, stream, d_major, d_minor, n, (unsigned)major_begin, (unsigned)major_end, keys_a);

    // stable LSD radix sort on the significant bits only: minor in [0, 32), major above (the all-ones key of dropped
    // triplets is all ones in those bits too, and no real key is: minor indices stay below 2^31)
    const unsigned end_bit = (unsigned)(32 + bits_for(dim_major + 1));
    size_t tmp_bytes = 0;
    TRY_OR_CLEAN(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a, keys_b, d_val, vals_b, n, 0u, end_bit, stream));
    TRY_OR_CLEAN(pmf_alloc(&tmp, tmp_bytes ? tmp_bytes : 16, stream));
    TRY_OR_CLEAN(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_a, keys_b, d_val, vals_b, n, 0u, end_bit, stream));
    pmf_free(tmp, stream); tmp = nullptr;

    // equal keys -> one entry holding the sum (keys_a is reused for the unique keys)
    size_t tmp2 = 0;
    TRY_OR_CLEAN(rocprim::reduce_by_key(nullptr, tmp2, keys_b, vals_b, (unsigned int)n, keys_a, out_val, d_count,
                                        rocprim::plus<real_t>(), rocprim::equal_to<unsigned long long>(), stream));
    TRY_OR_CLEAN(pmf_alloc(&tmp, tmp2 ? tmp2 : 16, stream));
    TRY_OR_CLEAN(rocprim::reduce_by_key(tmp, tmp2, keys_b, vals_b, (unsigned int)n, keys_a, out_val, d_count,
                                        rocprim::plus<real_t>(), rocprim::equal_to<unsigned long long>(), stream));
    size_t uniq = 0;
    TRY_OR_CLEAN(pmf_download(&uniq, d_count, sizeof(size_t), stream));
    if (uniq > 0) {   // the trailing entry of the dropped triplets, if any
        unsigned long long last = 0;
        TRY_OR_CLEAN(pmf_download(&last, keys_a + (uniq - 1), sizeof(last), stream));
        if (last == ~0ull) uniq--;
    }
    const unsigned g2 = (unsigned)std::min<size_t>((uniq + 255) / 256 + 1, 256 * 8);
    hipLaunchKernelGGL(split_keys, dim3(g2), dim3(256), 0, stream, keys_a, uniq, out_minor);
    const unsigned g3 = (unsigned)std::min<size_t>((dim_major + 256) / 256, 256 * 8);
    hipLaunchKernelGGL(row_pointers, dim3(g3), dim3(256), 0, stream, keys_a, uniq, dim_major, out_indptr);
    TRY_OR_CLEAN(hipGetLastError());
    TRY_OR_CLEAN(hipStreamSynchronize(stream));
    *nnz_out = uniq;
    cleanup();
#undef TRY_OR_CLEAN
    return 0;
}

// nloc consecutive rows (d_indptr points at the first one's CSR pointer; its row id is `base`) sorted by length,
// longest first, equal lengths in row order: d_perm[i] = row id, d_len_sorted[i] = its length.  (Stable LSD radix
// sort of (length, row) pairs, descending.)
int poismf_hip_device_sort_rows(const unsigned long long* d_indptr, size_t nloc, unsigned base, unsigned* d_perm, unsigned* d_len_sorted,
                                hipStream_t stream)
{
    if (nloc == 0) return 0;
    unsigned *len = nullptr, *ids = nullptr;
    void* tmp = nullptr;
    auto cleanup = [&]() {
        pmf_free(len, stream);
        pmf_free(ids, stream);
        pmf_free(tmp, stream);
    };
    if (pmf_alloc(&len, sizeof(unsigned) * nloc, stream) != hipSuccess || pmf_alloc(&ids, sizeof(unsigned) * nloc, stream) != hipSuccess) { cleanup(); return 1; }
    const unsigned grid = (unsigned)std::min<size_t>((nloc + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(row_len_kernel, dim3(grid), dim3(256), 0, stream, d_indptr, nloc, base, len, ids);
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs_desc(nullptr, bytes, len, d_len_sorted, ids, d_perm, nloc, 0u, 32u, stream) != hipSuccess ||
        pmf_alloc(&tmp, bytes ? bytes : 16, stream) != hipSuccess ||
        rocprim::radix_sort_pairs_desc(tmp, bytes, len, d_len_sorted, ids, d_perm, nloc, 0u, 32u, stream) != hipSuccess) { cleanup(); return 1; }
    cleanup();
    return 0;
}

// dst[i] = (u32) src[i] on the device (host CSR indices arrive as size_t)
int poismf_hip_device_narrow(const unsigned long long* d_src, size_t n, unsigned* d_dst, hipStream_t stream)
{
    if (n == 0) return 0;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(narrow_kernel, dim3(grid), dim3(256), 0, stream, d_src, n, d_dst);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" {

int poismf_hip_coo_to_csr_csc(const sparse_ix* row, const sparse_ix* col, const real_t* val, size_t n, size_t dimA, size_t dimB,
                              real_t* csr_val, sparse_ix* csr_indices, sparse_ix* csr_indptr, real_t* csc_val,
                              sparse_ix* csc_indices, sparse_ix* csc_indptr, size_t* nnz_out)
{
    if (n == 0 || n > 0xffffffffull || dimA > 0x7fffffffull || dimB > 0x7fffffffull) return 1;
    int device = 0;
    if (const char* e = getenv("POISMF_HIP_DEVICE")) device = atoi(e);
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = nullptr;
    unsigned *d_row = nullptr, *d_col = nullptr, *d_minor = nullptr;
    real_t *d_val = nullptr, *d_oval = nullptr;
    unsigned long long* d_ptr = nullptr;
    std::vector<unsigned> h32(n);
    std::vector<unsigned long long> hptr(std::max(dimA, dimB) + 1);
    int rc = 1;
    do {
        if (pmf_malloc_retry((void**)&d_row, sizeof(unsigned) * n) != hipSuccess || pmf_malloc_retry((void**)&d_col, sizeof(unsigned) * n) != hipSuccess ||
            pmf_malloc_retry((void**)&d_minor, sizeof(unsigned) * n) != hipSuccess || pmf_malloc_retry((void**)&d_val, sizeof(real_t) * n) != hipSuccess ||
            pmf_malloc_retry((void**)&d_oval, sizeof(real_t) * n) != hipSuccess ||
            pmf_malloc_retry((void**)&d_ptr, sizeof(unsigned long long) * (std::max(dimA, dimB) + 1)) != hipSuccess)
            break;
        for (size_t i = 0; i < n; i++) h32[i] = (unsigned)row[i];
        if (hipMemcpy(d_row, h32.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice) != hipSuccess) break;
        for (size_t i = 0; i < n; i++) h32[i] = (unsigned)col[i];
        if (hipMemcpy(d_col, h32.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_val, val, sizeof(real_t) * n, hipMemcpyHostToDevice) != hipSuccess) break;
        bool ok = true;
        for (int pass = 0; pass < 2 && ok; pass++) {
            const bool csr = pass == 0;
            size_t uniq = 0;
            if (poismf_hip_device_coo_to_cs(csr ? d_row : d_col, csr ? d_col : d_row, d_val, n, 0, csr ? dimA : dimB,
                                            d_minor, d_oval, d_ptr, &uniq, stream)) { ok = false; break; }
            const size_t dim = csr ? dimA : dimB;
            real_t* oval = csr ? csr_val : csc_val;
            sparse_ix* oidx = csr ? csr_indices : csc_indices;
            sparse_ix* optr = csr ? csr_indptr : csc_indptr;
            if (hipMemcpy(oval, d_oval, sizeof(real_t) * uniq, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(h32.data(), d_minor, sizeof(unsigned) * uniq, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(hptr.data(), d_ptr, sizeof(unsigned long long) * (dim + 1), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
            for (size_t i = 0; i < uniq; i++) oidx[i] = (sparse_ix)h32[i];
            for (size_t i = 0; i <= dim; i++) optr[i] = (sparse_ix)hptr[i];
            *nnz_out = uniq;
        }
        if (ok) rc = 0;
    } while (0);
    if (d_row) (void)hipFree(d_row);
    if (d_col) (void)hipFree(d_col);
    if (d_minor) (void)hipFree(d_minor);
    if (d_val) (void)hipFree(d_val);
    if (d_oval) (void)hipFree(d_oval);
    if (d_ptr) (void)hipFree(d_ptr);
    if (rc) fprintf(stderr, "Error: out of memory.\n");
    return rc;
}

}  // extern "C"
