// poismf_hip.hip -- kernels and host side of the MI355X implementation of poismf's alternating
// factor-update path.  Compiled twice (real_t = double, and float with -DUSE_FLOAT) into
// libpoismf_hip_d.so / libpoismf_hip_f.so; the C-ABI is declared in include/poismf_hip.h.
//
// Host (this file, plain C++): the outer A/B alternation, step schedule, early-stop logic, SIGINT
// plumbing and return codes of run_poismf (ref: src/poismf.c:435-632), plus a device-resident
// session used by bench.py and by the one-process-per-GPU driver.
// Device: one launch per (half-sweep, row bin); one wavefront per row (row_eval.hpp, solvers.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/poismf_hip.h"
#include "reg_eval.hpp"
#include "solvers.hpp"

using namespace pmf;

// =================================================================================================
// Kernels
// =================================================================================================
struct RowDesc;
template <class T> struct HalfArgs {
    T* M;                             // factor being updated, [dimM x k]
    T* Mp;                            // its line-padded copy (row stride ldM), or nullptr: updated rows go to both
    int ldM;
    const T* F;                       // opposing factor, [dimF x k] (+16 B of slack)
    const unsigned long long* indptr; // shard-local CSR row pointers (nrows_local + 1)
    const unsigned* indices;
    const T* values;
    const unsigned* perm;             // shard-local row ids, sorted by nnz descending
    const struct RowDesc* desc;       // the same order, with each row's CSR offset and length (one load per row)
    unsigned perm_begin, nrows;       // this launch covers perm[perm_begin, perm_begin + nrows)
    unsigned row_offset;              // first global row of the shard (M row = row_offset + local id)
    const T* bsum;                    // k-vector: colsum(F) + l1 (pre-scaled for PG when w == 1)
    TileGeom geom;
    RowParams<T> P;
    int reuse_prev, early_stop;
    unsigned* n_unchanged;
    unsigned* queue;                  // != nullptr: rows are handed out dynamically through this counter
    unsigned* eval_rows;              // != nullptr (profiling sessions): [local row] += passes over that row's tile
};

enum { K_PG = 3, K_CG = 2, K_TNCG = 1 };

// Translation units.  The row kernels of the three solvers are the bulk of the compile time, so the build compiles this
// file four times per precision (poismf_amd/build.py): PMF_TU = 0 is the host side (sessions, run_poismf, planning; no
// row kernel), PMF_TU = K_TNCG / K_CG / K_PG instantiate the row kernels of one solver each and export one function,
// pmf_launch_one_tu<N>.  Without PMF_TU everything lands in one translation unit.
#ifndef PMF_TU
#define PMF_TU -1
#endif
constexpr bool tu_has(int kmethod) { return PMF_TU == -1 || PMF_TU == kmethod; }
#define PMF_TU_HOST (PMF_TU <= 0)
#define PMF_TU_KERNELS (PMF_TU != 0)

// One row of the sorted order: where its nonzeros start in the shard's CSR arrays, how many, and which row it is.
struct RowDesc { unsigned p0_lo, p0_hi, nnz, lrow; };

// Everything after the row's tile has been requested: starting point, per-row constant term, inner solver, store.
template <class EV, class T, int NC, int METHOD>
__device__ __forceinline__ void solve_row(const HalfArgs<T>& a, EV& ev, const T (&bs)[NC], unsigned lrow, unsigned nnz)
{
    const int k = a.geom.k;
    T* out = a.M + (size_t)(a.row_offset + lrow) * (size_t)k;
    T* out_p = a.Mp != nullptr ? a.Mp + (size_t)(a.row_offset + lrow) * (size_t)a.ldM : nullptr;
    T x[NC];
    if (nnz == 0) {  // rows without data are forced to zero every half (quirk Q7)
        PMF_EW x[i] = (T)0;
        ev.store_vec(out, x);
        if (out_p != nullptr) ev.store_vec(out_p, x);
        return;
    }
    ev.load_vec(out, x);
    ev.n_eval = 0;

    // per-row constant term: the k-vector itself, or (w != 1) the reference's Bsum_w row
    //   (w - 1) sum_j F_j + Bsum          ref: src/poismf.c:85-123 (adjustment_Bsum)
    T shift[NC];
    PMF_EW shift[i] = bs[i];
    const bool weighted = a.P.w != (T)1;
    if (weighted) {
        T cs[NC];
        PMF_EW cs[i] = (T)0;
        ev.tile_colsum(cs);
        const T wm1 = a.P.w - (T)1.;
        PMF_EW {
            shift[i] = cs[i] * wm1;
            shift[i] = shift[i] + bs[i];
        }
        if (METHOD == K_PG) { PMF_EW shift[i] = shift[i] * a.P.neg_step; }  // dscal_large, ref: :526, :576
    }

    if constexpr (METHOD == K_PG) {
        pg_row(ev, a.P, x, shift);
    } else if constexpr (METHOD == K_CG) {
        if (a.P.limit_step && ev.pq_cap > 0 && nnz <= (unsigned)ev.pq_cap) cg_row_cached(ev, a.P, shift, x, weighted);  // streamed rows only, see plan_geom
        else cg_row(ev, a.P, shift, x, weighted);
    } else {
        T prev[NC];
        PMF_EW prev[i] = x[i];
        if (!a.reuse_prev) { PMF_EW x[i] = (T)1e-3; }                   // ref: src/poismf.c:379-381
        (void)Tnc<T, NC, EV>::minimize(ev, a.P, shift, x);
        if (a.early_stop) {                                             // ref: src/poismf.c:393-396
            PMF_EW prev[i] = prev[i] - x[i];
            const T moved = ev.dot(prev, prev);
            if ((double)moved <= 1e-4 && ev.lane == 0 && ev.wid == 0) atomicAdd(a.n_unchanged, 1u);
        }
    }
    ev.store_vec(out, x);
    if (out_p != nullptr) ev.store_vec(out_p, x);
    // per-row evaluation counts for the pass-weighted effective traffic (a plain read-modify-write: the row
    // has one owner; a shared counter here costs 4x the kernel time in contention)
    if (a.eval_rows != nullptr && ev.lane == 0 && ev.wid == 0) a.eval_rows[lrow] += ev.n_eval;
}

// A wavefront (or, NW > 1, a workgroup of NW wavefronts) walks rows blockIdx.x, blockIdx.x + gridDim.x, ... of the
// nnz-sorted permutation, or pulls them from a device-wide queue.
//
// Row hand-out.  PG does the same work for every nonzero, so the nnz-sorted rows are dealt out statically
// (row r, r + grid, ...).  CG and TNCG take anything from a handful to ~400 evaluations per row: there rows are
// pulled from a device-wide counter in nnz-descending order (longest first -- the GPU form of the reference's
// `schedule(dynamic)`, ref: src/poismf.c:296, :352), one returning atomic per row.
template <class EV, class T, int NC, int METHOD, int NW>
__device__ __forceinline__ void sweep_rows(const HalfArgs<T>& a, EV& ev, unsigned char* smem)
{
    ev.init(a.geom, a.F, smem);
    T bs[NC];
    ev.load_vec(a.bsum, bs);
    const RowDesc* desc = a.desc + a.perm_begin;

    if constexpr (EV::PIPELINED) {
        // Software pipeline over the rows of this wave.  A row costs three dependent round trips to memory -- its
        // descriptor, its indices, the factor rows those name -- and the solver in between leaves the memory pipe idle.
        // Tickets (and descriptors) are fetched two rows ahead and the indices one row ahead, so that a row's gather
        // starts the moment the previous row is stored.
        unsigned r = blockIdx.x;
        auto ticket = [&]() -> unsigned {
            if (a.queue != nullptr) {
                if constexpr (NW > 1) {   // one atomic per workgroup, broadcast through LDS
                    unsigned* slot = ev.ticket_slot();
                    if (threadIdx.x == 0) *slot = atomicAdd(a.queue, 1u);
                    __syncthreads();
                    const unsigned t = uniform(*slot);
                    __syncthreads();
                    return t;
                }
                unsigned t = 0;
                if (ev.lane == 0) t = atomicAdd(a.queue, 1u);
                return uniform(t);
            }
            const unsigned t = r;
            r += gridDim.x;
            return t;
        };
        auto fetch = [&](unsigned t) -> RowDesc { return desc[t < a.nrows ? t : 0u]; };
        unsigned t0 = ticket(), t1 = ticket();
        RowDesc d0 = fetch(t0), d1 = fetch(t1);
        if (t0 < a.nrows) ev.fetch_meta(a.indices + (((unsigned long long)d0.p0_hi << 32) | d0.p0_lo), d0.nnz);
        while (t0 < a.nrows) {
            const unsigned long long p0 = ((unsigned long long)d0.p0_hi << 32) | d0.p0_lo;
            if (d0.nnz != 0) ev.gather(a.values + p0, d0.nnz);                  // indices are here: request the tile
            const unsigned t2 = ticket();
            const RowDesc d2 = fetch(t2);
            if (t1 < a.nrows) ev.fetch_meta(a.indices + (((unsigned long long)d1.p0_hi << 32) | d1.p0_lo), d1.nnz);
            solve_row<EV, T, NC, METHOD>(a, ev, bs, d0.lrow, d0.nnz);
            t0 = t1; d0 = d1;
            t1 = t2; d1 = d2;
        }
        return;
    } else {
        unsigned r = blockIdx.x;
        for (;;) {
            if (a.queue != nullptr) {
                if constexpr (NW > 1) {
                    // broadcast through the last 16 bytes of the DYNAMIC LDS block (a static __shared__ object
                    // would shift the dynamic base off 16-byte alignment and slow every ds_read_b128 down)
                    unsigned* next_row = ev.ticket_slot();
                    if (threadIdx.x == 0) *next_row = atomicAdd(a.queue, 1u);
                    __syncthreads();
                    r = uniform(*next_row);
                    __syncthreads();
                } else {
                    unsigned t = 0;
                    if (ev.lane == 0) t = atomicAdd(a.queue, 1u);
                    r = uniform(t);
                }
            }
            if (r >= a.nrows) break;
            const RowDesc d = desc[r];
            r += gridDim.x;
            const unsigned nnz = uniform(d.nnz);
            const unsigned long long p0 = ((unsigned long long)uniform(d.p0_hi) << 32) | uniform(d.p0_lo);
            if (nnz != 0) ev.begin_row(a.indices + p0, a.values + p0, nnz);
            solve_row<EV, T, NC, METHOD>(a, ev, bs, uniform(d.lrow), nnz);
        }
    }
}

#ifndef PMF_TNC_PREFETCH
#define PMF_TNC_PREFETCH 0
#endif
// LDS-tile engine (row_eval.hpp): one wavefront (= one 64-thread workgroup, so no workgroup barrier is ever needed and
// the LDS tile is private), or NW wavefronts per row for the long-row path.
template <class T, int NC, int METHOD, int SL, int NW>
__global__ __launch_bounds__(WAVE* NW) void half_sweep_kernel(const HalfArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // streamed rows prefetch the next chunk's tile (row_eval.hpp, PF); TNC's register budget is spent already
#ifdef PMF_FORCE_NO_PF
    constexpr bool PF = false;
#else
    // (only in the instances with a compile-time slot count: the generic fp64 CG instance is at 512 registers already,
    // and the 14 slots in flight pushed it into scratch -- and into wrong results on the k = 200 test)
    constexpr bool PF = SL > 0 && (METHOD != K_TNCG || PMF_TNC_PREFETCH);
#endif
    RowEval<T, NC, SL, NW, PF> ev;
#ifdef PMF_TIMING
    const unsigned long long t_kernel = __builtin_amdgcn_s_memtime();
#endif
    sweep_rows<RowEval<T, NC, SL, NW, PF>, T, NC, METHOD, NW>(a, ev, smem);
#ifdef PMF_TIMING
    ev.tacc[5] = __builtin_amdgcn_s_memtime() - t_kernel;
    if (ev.lane == 0 && ev.wid == 0)
        for (int q = 0; q < 6; q++) atomicAdd(&g_pmf_timing[q], ev.tacc[q]);
#endif
}

// Waves per SIMD the register allocator is asked to make room for: the largest count whose VGPR budget (512 per SIMD
// lane, granules of 8) holds the tile (TR = 4 S NS registers) plus an allowance for the solver.  The allowances are
// tuned on C2 (ms per sweep): PG 40 / 56 / 72 -> 1.32 / 1.15 / 1.31; CG 60 / 90 / 120 / 150 -> 3.68 / 3.38 / 3.31 / 3.53;
// TNCG 30 / 60 / 80 / 100 / 150 / 200 -> 57 / 26 / 17.5 / 16.6 / 18.4 / 20.2 (TNC keeps ~21 k-vectors: below its real
// need the idle ones spill around the evaluations, which is cheaper than giving up a wave -- up to a point).
#ifndef PMF_PG_EXTRA
#define PMF_PG_EXTRA 56
#endif
#ifndef PMF_CG_EXTRA
#define PMF_CG_EXTRA 120
#endif
#ifndef PMF_TNC_EXTRA
#define PMF_TNC_EXTRA 100
#endif
constexpr int reg_waves(int tile_regs, int method)
{
    const int need = tile_regs + (method == K_PG ? PMF_PG_EXTRA : method == K_CG ? PMF_CG_EXTRA : PMF_TNC_EXTRA);
    for (int w : { 8, 6, 5, 4, 3, 2 })
        if ((512 / w) / 8 * 8 >= need) return w;
    return 1;
}
// Register-tile engine (reg_eval.hpp) for rows of at most (64 / G) S nonzeros: no LDS at all, waves per CU set by VGPRs.
template <class T, int METHOD, int S, int G, int NS>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(reg_waves(4 * S * NS, METHOD)))) void half_sweep_reg_kernel(const HalfArgs<T> a)
{
    RegEval<T, S, G, NS> ev;
#ifdef PMF_TIMING
    const unsigned long long t_kernel = __builtin_amdgcn_s_memtime();
#endif
    sweep_rows<RegEval<T, S, G, NS>, T, RegEval<T, S, G, NS>::NC, METHOD, 1>(a, ev, nullptr);
#ifdef PMF_TIMING
    if (ev.lane == 0) atomicAdd(&g_pmf_timing[5], __builtin_amdgcn_s_memtime() - t_kernel);
#endif
}

// The same engine with NW = 2, 4 or 8 wavefronts per row (reg_eval.hpp, NW_ > 1): rows of up to NW (64 / G) S nonzeros.
// The fewest waves whose shares fit their registers are used: every evaluation ends in a barrier and a round trip
// through LDS, which costs more the more waves take part (C2-shaped PG(10), ns per nonzero and half: 100-nonzero rows on
// one wave 0.06; 200-nonzero rows 0.137 on eight waves).
constexpr int REG_NW_MAX = 8;
constexpr int regw_waves(int tile_regs, int method, int nw)
{
    const int w = reg_waves(tile_regs, method);
    return nw >= 8 && w < 2 ? 2 : w;   // eight waves are two per SIMD
}
template <class T, int METHOD, int S, int G, int NS, int NW>
__global__ __launch_bounds__(WAVE* NW) __attribute__((amdgpu_waves_per_eu(regw_waves(4 * S * NS, METHOD, NW)))) void half_sweep_regw_kernel(const HalfArgs<T> a)
{
    using EV = RegEval<T, S, G, NS, NW>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[EV::SMEM_BYTES];
    EV ev;
    sweep_rows<EV, T, EV::NC, METHOD, NW>(a, ev, smem);
}

#if PMF_TU_HOST
// ---- self-test of wave_ops.hpp's d_log against the device library's log --------------------------------------------
__global__ __launch_bounds__(256) void selftest_log_kernel(unsigned long long n, unsigned long long* worst_ulp, unsigned* mismatched_specials)
{
    unsigned long long worst = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        // arguments: a dense sweep of [1/4, 4] (where cancellation is worst), then 2^-1074 .. 2^1023 through bit patterns
        double x;
        if (i < n / 2) x = 0.25 + 3.75 * (double)i / (double)(n / 2);
        else {
            const unsigned long long j = i - n / 2, m = n - n / 2;
            const unsigned long long bits = (unsigned long long)((double)j / (double)m * (double)0x7fefffffffffffffULL);
            x = __builtin_bit_cast(double, bits ? bits : 1ULL);
        }
        const double a = d_log(x), b = d_log_lib(x);
        const long long ia = __builtin_bit_cast(long long, a), ib = __builtin_bit_cast(long long, b);
        const unsigned long long d = (unsigned long long)(ia > ib ? ia - ib : ib - ia);   // same sign: distance in ulps
        if ((ia < 0) == (ib < 0)) worst = d > worst ? d : worst;
        else if (a != b) worst = ~0ULL >> 1;
    }
    atomicMax(worst_ulp, worst);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double sp[6] = { 0.0, -0.0, -1.0, __builtin_inf(), -__builtin_inf(), __builtin_nan("") };
        unsigned bad = 0;
        for (int q = 0; q < 6; q++) {
            const double a = d_log(sp[q]), b = d_log_lib(sp[q]);
            const bool same = (a != a && b != b) || a == b;
            bad += same ? 0u : 1u;
        }
        *mismatched_specials = bad;
    }
}

#endif  // PMF_TU_HOST

// ---- compact factor -> line-padded copy (the pad columns stay zero from the allocation) --------------------------
template <class T> __global__ __launch_bounds__(256) void repad_kernel(const T* src, T* dst, size_t n, int k, int ld)
{
    const size_t total = n * (size_t)k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (size_t)k;
        const int c = (int)(i - r * (size_t)k);
        dst[r * (size_t)ld + c] = src[i];
    }
}

// ---- column sums of a dense [n x k] factor: sum_by_cols, ref: src/poismf.c:77-83 ---------------------
// stage 1: wave w accumulates rows w, w + nw, ...; stage 2: one wave adds the nw partials in order,
// then applies `+ l1` (ref: :513-514) and the PG pre-scalings (ref: :523-526, :573-577, quirk Q1).
// stage 1: blocks of 8 waves; wave w of block b accumulates rows (b*8 + w), + 8*grid, ...; the 8 wave totals are added
// through LDS in wave order, so the 256-way partial written by the block is bit-reproducible.
constexpr int COLSUM_BLOCK_WAVES = 8;
template <class T, int NC>
__global__ __launch_bounds__(WAVE* COLSUM_BLOCK_WAVES) void colsum_partial_kernel(const T* M, size_t n, int k, T* partial)
{
    __shared__ T part[COLSUM_BLOCK_WAVES][NC * WAVE];
    const int lane = lane_id();
    const int w = (int)(threadIdx.x / WAVE);
    T acc[NC];
    PMF_EW acc[i] = (T)0;
    // four rows in flight per wave (the loop is latency-bound: one 200-byte row per trip); added in row order as before
    const size_t stride = (size_t)gridDim.x * COLSUM_BLOCK_WAVES;
    size_t r = (size_t)blockIdx.x * COLSUM_BLOCK_WAVES + w;
    for (; r + 3 * stride < n; r += 4 * stride) {
        T v[4][NC];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const T* row = M + (r + u * stride) * (size_t)k;
            PMF_EW v[u][i] = (lane + WAVE * i < k) ? row[lane + WAVE * i] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { PMF_EW acc[i] += v[u][i]; }
    }
    for (; r < n; r += stride) {
        const T* row = M + r * (size_t)k;
        PMF_EW if (lane + WAVE * i < k) acc[i] += row[lane + WAVE * i];
    }
    PMF_EW part[w][lane + WAVE * i] = acc[i];
    __syncthreads();
    if (w == 0) {
        PMF_EW {
            const int c = lane + WAVE * i;
            if (c < k) {
                T s = part[0][c];
                for (int q = 1; q < COLSUM_BLOCK_WAVES; q++) s += part[q][c];
                partial[(size_t)blockIdx.x * k + c] = s;
            }
        }
    }
}
// stage 2: 16 waves; wave w adds partials w, w + 16, ... in order, then wave 0 adds the 16 wave totals in order
// (fixed summation order => bit-reproducible), applies `+ l1` and the PG pre-scalings.
constexpr int COLSUM_FINAL_WAVES = 16;
template <class T, int NC>
__global__ __launch_bounds__(WAVE* COLSUM_FINAL_WAVES) void colsum_final_kernel(const T* partial, int nw, int k, T l1, T scale,
                                                                                 int nscale, T* out)
{
    __shared__ T part[COLSUM_FINAL_WAVES][NC * WAVE];
    const int lane = lane_id();
    const int w = (int)(threadIdx.x / WAVE);
    T acc[NC];
    PMF_EW acc[i] = (T)0;
    int r = w;
    for (; r + 7 * COLSUM_FINAL_WAVES < nw; r += 8 * COLSUM_FINAL_WAVES) {   // eight loads in flight, same order of adds
        T v[8][NC];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            PMF_EW v[u][i] = (lane + WAVE * i < k) ? partial[(size_t)(r + u * COLSUM_FINAL_WAVES) * k + lane + WAVE * i] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { PMF_EW acc[i] += v[u][i]; }
    }
    for (; r < nw; r += COLSUM_FINAL_WAVES) {
        PMF_EW if (lane + WAVE * i < k) acc[i] += partial[(size_t)r * k + lane + WAVE * i];
    }
    PMF_EW part[w][lane + WAVE * i] = acc[i];
    __syncthreads();
    if (w == 0) {
        PMF_EW {
            const int c = lane + WAVE * i;
            if (c < k) {
                T s = part[0][c];
                for (int q = 1; q < COLSUM_FINAL_WAVES; q++) s += part[q][c];
                if (l1 > (T)0.) s += l1;
                for (int q = 0; q < nscale; q++) s *= scale;
                out[c] = s;
            }
        }
    }
}

// =================================================================================================
// Host side
// =================================================================================================
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "poismf_hip: %s failed: %s\n", #expr, hipGetErrorString(e_));          \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

namespace {

constexpr size_t LDS_PER_CU = 160 * 1024;
constexpr size_t LDS_RESIDENT_LIMIT = 64 * 1024;  // largest tile a single wave may claim
constexpr int NUM_CU = 256;
constexpr int MAX_LAUNCHES = 64;  // row bins per half-sweep (16 fine classes + powers of two up to 2^31)

struct Bin {
    unsigned begin, count;  // range of the nnz-sorted permutation
    unsigned max_nnz;       // longest row actually in the bin (sizes tiles)
    unsigned cls;           // upper bound of the bin's length class (decides the code path: a function of the row alone)
};

struct Half {
    size_t dimM = 0, dimF = 0;
    size_t row_begin = 0, row_end = 0;  // shard
    size_t nnz = 0;
    unsigned long long* d_indptr = nullptr;
    unsigned* d_indices = nullptr;
    real_t* d_values = nullptr;
    unsigned* d_perm = nullptr;
    RowDesc* d_desc = nullptr;
    unsigned* d_eval_rows = nullptr;      // per local row: passes over its tile while profiling (allocated on demand)
    std::vector<unsigned> row_nnz;        // host copy of the row lengths (for the pass-weighted traffic report)
    std::vector<Bin> bins;
};

struct ProfRec { hipEvent_t t0, t1; int which; };

}  // namespace

struct poismf_hip_session {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipStream_t aux_stream = nullptr;  // long-row launches run here, next to the other bins
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    size_t dimA = 0, dimB = 0, k = 0;
    real_t *dA = nullptr, *dB = nullptr;
    // line-padded copies of the factors for the gathers (row stride `ld` elements = a multiple of 128 bytes), kept when
    // that cuts the bytes a gathered row drags in by >= 10 %; refreshed from the compact factor before each half-sweep
    real_t *dAp = nullptr, *dBp = nullptr;
    size_t ld = 0;
    bool padded_fresh[2] = { false, false };  // [0]: dBp mirrors dB, [1]: dAp mirrors dA
    Half half[2];  // [0]: rows of B (CSC), [1]: rows of A (CSR)
    real_t* d_bsum = nullptr;
    real_t* d_partial = nullptr;
    unsigned* d_counter = nullptr;
    unsigned* d_queue = nullptr;      // one row-queue head per launch of a half-sweep
    int colsum_waves = 512;           // blocks (of 8 waves) in the first stage of the column sums
    bool profiling = false;
    std::vector<ProfRec> prof;
};

namespace {

void free_half(Half& h)
{
    if (h.d_indptr) (void)hipFree(h.d_indptr);
    if (h.d_indices) (void)hipFree(h.d_indices);
    if (h.d_values) (void)hipFree(h.d_values);
    if (h.d_perm) (void)hipFree(h.d_perm);
    if (h.d_desc) (void)hipFree(h.d_desc);
    if (h.d_eval_rows) (void)hipFree(h.d_eval_rows);
    h = Half();
}

// Upload rows [r0, r1) of a host CSR (size_t indices) with shard-local pointers and u32 indices; build
// the nnz-descending permutation and its power-of-two bins.
int build_half(Half& h, hipStream_t stream, const real_t* val, const sparse_ix* indptr, const sparse_ix* indices,
               size_t dimM, size_t dimF, size_t r0, size_t r1)
{
    h.dimM = dimM; h.dimF = dimF; h.row_begin = r0; h.row_end = r1;
    const size_t nloc = r1 - r0;
    const size_t base = indptr[r0];
    h.nnz = indptr[r1] - base;
    std::vector<unsigned long long> lptr(nloc + 1);
    for (size_t i = 0; i <= nloc; i++) lptr[i] = (unsigned long long)(indptr[r0 + i] - base);
    std::vector<unsigned> lidx(h.nnz ? h.nnz : 1);
    for (size_t i = 0; i < h.nnz; i++) lidx[i] = (unsigned)indices[base + i];

    // counting sort of rows by nnz, descending (stable: ties keep row order)
    std::vector<unsigned> perm(nloc ? nloc : 1);
    {
        std::vector<std::pair<unsigned, unsigned>> key(nloc);
        for (size_t i = 0; i < nloc; i++) key[i] = { (unsigned)(lptr[i + 1] - lptr[i]), (unsigned)i };
        std::stable_sort(key.begin(), key.end(),
                         [](const std::pair<unsigned, unsigned>& a, const std::pair<unsigned, unsigned>& b) { return a.first > b.first; });
        h.bins.clear();
        for (size_t i = 0; i < nloc; i++) {
            perm[i] = key[i].second;
            const unsigned n = key[i].first;
            // bin classes: multiples of 16 up to 256 nonzeros, multiples of 64 up to 1280 (the hand-overs between 1, 2, 4
            // and 8 waves per row fall on those), then powers of two.  The LDS tile of a launch is sized by the longest
            // row of its bin, and LDS is what limits the waves per CU, so fine classes where most rows live buy
            // occupancy (C2: 100 +- 10 nnz per row -> 7 waves per CU instead of 5).  Which ENGINE a row takes, and how
            // many waves share it, is decided by the class bound alone -- never by which other rows happen to be in the
            // shard -- so a row's arithmetic does not depend on how the matrix is cut into shards.
            unsigned cls;
            if (n <= 256) cls = std::max(16u, (n + 15u) / 16u * 16u);
            else if (n <= 1280) cls = (n + 63u) / 64u * 64u;
            else { cls = 2048; while (cls < n) cls <<= 1; }
            if (h.bins.empty() || cls != h.bins.back().cls) h.bins.push_back({ (unsigned)i, 0u, 0u, cls });
            h.bins.back().count++;
        }
        // record the true maximum of each bin (first row, since sorted)
        for (auto& b : h.bins) b.max_nnz = key[b.begin].first;
    }
    HIP_TRY(hipMalloc(&h.d_indptr, sizeof(unsigned long long) * (nloc + 1)));
    HIP_TRY(hipMalloc(&h.d_indices, sizeof(unsigned) * (h.nnz ? h.nnz : 1)));
    HIP_TRY(hipMalloc(&h.d_values, sizeof(real_t) * (h.nnz ? h.nnz : 1)));
    HIP_TRY(hipMalloc(&h.d_perm, sizeof(unsigned) * (nloc ? nloc : 1)));
    HIP_TRY(hipMemcpyAsync(h.d_indptr, lptr.data(), sizeof(unsigned long long) * (nloc + 1), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(h.d_indices, lidx.data(), sizeof(unsigned) * h.nnz, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(h.d_values, val + base, sizeof(real_t) * h.nnz, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(h.d_perm, perm.data(), sizeof(unsigned) * nloc, hipMemcpyHostToDevice, stream));
    h.row_nnz.resize(nloc);
    for (size_t i = 0; i < nloc; i++) h.row_nnz[i] = (unsigned)(lptr[i + 1] - lptr[i]);
    std::vector<RowDesc> desc(nloc ? nloc : 1);
    for (size_t i = 0; i < nloc; i++) {
        const unsigned long long p0 = lptr[perm[i]];
        desc[i] = { (unsigned)p0, (unsigned)(p0 >> 32), (unsigned)(lptr[perm[i] + 1] - p0), perm[i] };
    }
    HIP_TRY(hipMalloc(&h.d_desc, sizeof(RowDesc) * (nloc ? nloc : 1)));
    HIP_TRY(hipMemcpyAsync(h.d_desc, desc.data(), sizeof(RowDesc) * nloc, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));  // host staging vectors die here
    return 0;
}

// Slot counts with a compile-time specialisation: the k values of the BASELINE configs
// (fp32: k = 49..52 -> 13 slots, k = 97..100 -> 25; fp64: k = 49..50 -> 25, k = 99..100 -> 50).
#ifdef USE_FLOAT
constexpr int SPECIAL_SL_A = 13, SPECIAL_SL_B = 25;
#else
constexpr int SPECIAL_SL_A = 25, SPECIAL_SL_B = 50;
#endif

bool prefetch_enabled()
{
    static const bool off = getenv("POISMF_HIP_NO_PREFETCH") != nullptr;  // testing knob
    return !off;
}

TileGeom plan_geom(size_t k, unsigned bin_max_nnz, bool single_pass, bool want_pq)
{
    TileGeom g;
    g.k = (int)k;
    g.pq_cap = 0;
    g.prefetch = 0;
    g.zero_row = 0;
    g.ldF = (int)k;
    g.s_load = (int)((k * sizeof(real_t) + 15) / 16);
    g.s_stride = g.s_load | 1;
    g.group = g.s_load <= 16 ? 16 : (g.s_load <= 32 ? 32 : 64);
    // tile capacity: whole rows of this bin if that fits the per-wave budget, else stream in chunks.
    // A single-pass solver (PG with one update) gains nothing from residency: keep tiles small there
    // so that many waves per CU keep gathers in flight.
    const unsigned want = std::max(16u, bin_max_nnz);
    // chunk for streamed rows: ~14 KiB of tile per wave keeps >= 10 waves per CU gathering (measured on C2:
    // 128-nonzero chunks 1.46 ms per sweep, 64 or 32: 0.80-0.85 ms)
    unsigned stream_chunk = (unsigned)(14336 / ((size_t)g.s_stride * 16)) / 16 * 16;
    stream_chunk = std::min(128u, std::max(16u, stream_chunk));
    // multi-pass solvers: at least 32 nonzeros per chunk while four such tiles still fit a CU's LDS next to everything
    // else (C5, TNCG fp64 k = 100: 16 -> 32 nonzeros takes the B half from 1091 to 956 ms; 48: 1051)
    if (!single_pass && (size_t)32 * g.s_stride * 16 <= 28 * 1024) stream_chunk = std::max(stream_chunk, 32u);
    // with the next chunk's tile requested a chunk ahead (row_eval.hpp, PF) a larger chunk amortises the per-chunk
    // overhead without exposing its gather: as many nonzeros as the PMF_PRE slots per lane in flight hold
    // (C3 B half, CG fp64: 32 nonzeros without prefetch 103.9 ms, with 96.6; 48 with prefetch 82.6; 64 without 111.7)
    if (!single_pass && prefetch_enabled() && ((size_t)g.s_load == (size_t)SPECIAL_SL_A || (size_t)g.s_load == (size_t)SPECIAL_SL_B))
        stream_chunk = std::max(stream_chunk, std::max(16u, (unsigned)(PMF_PRE * WAVE / g.s_load) / 16 * 16));
    if (const char* e = getenv("POISMF_HIP_STREAM_CHUNK")) stream_chunk = (unsigned)std::max(16, atoi(e));  // tuning knob
    unsigned cap = want;
    g.resident = 1;
    TileGeom probe = g;
    probe.cap = (int)cap;
    if (lds_bytes_per_wave(probe, sizeof(real_t)) > LDS_RESIDENT_LIMIT || (single_pass && cap > stream_chunk)) {
        cap = stream_chunk;
        probe.cap = (int)cap;
        while (cap > 16 && lds_bytes_per_wave(probe, sizeof(real_t)) > LDS_RESIDENT_LIMIT) { cap /= 2; probe.cap = (int)cap; }
        g.resident = 0;
    }
    g.cap = (int)cap;
    // CG: cache T.x and T.d per nonzero when both fit next to the tile (up to 48 KiB for the pair); longer rows
    // fall back to the direct line search
    static const bool no_cache = getenv("POISMF_HIP_CG_NOCACHE") != nullptr;  // testing knob
    // Only for streamed rows: there every line-search trial would otherwise be a fresh gather from L2/HBM
    // (C3 B half: 247 -> 146 ms).  For LDS-resident rows a trial is a cheap pass over the tile already and the
    // cached variant buys nothing: it halves the passes over the tile (C2 CG fp64: 20 -> 10 per row) and the sweep takes
    // the same 10.3 ms -- those rows are bound by the solver's chain of k-vector reductions and scalar decisions at one
    // wave per SIMD, not by the tile passes (POISMF_HIP_CG_CACHE_RESIDENT=1 switches it on for them).
    static const bool cache_resident = getenv("POISMF_HIP_CG_CACHE_RESIDENT") != nullptr;  // tuning knob
    if (want_pq && !no_cache && (!g.resident || cache_resident) && (size_t)2 * bin_max_nnz * sizeof(real_t) <= 48 * 1024)
        g.pq_cap = (int)((bin_max_nnz + 15u) / 16u * 16u);
    g.prefetch = (!g.resident && prefetch_enabled()) ? 1 : 0;
    return g;
}

template <int NC, int METHOD, int SL, int NW> int launch_bin(hipStream_t stream, const HalfArgs<real_t>& a, size_t lds, unsigned grid)
{
    auto kern = half_sweep_kernel<real_t, NC, METHOD, SL, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)LDS_PER_CU));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * NW), lds, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NC, int SL, int NW = 1> int launch_method(hipStream_t stream, int method, const HalfArgs<real_t>& a, size_t lds, unsigned grid)
{
    switch (method) {
        case POISMF_PG:
            if constexpr (tu_has(K_PG)) return launch_bin<NC, K_PG, SL, NW>(stream, a, lds, grid);
            else return 1;
        case POISMF_CG:
            if constexpr (tu_has(K_CG)) return launch_bin<NC, K_CG, SL, NW>(stream, a, lds, grid);
            else return 1;
        default:
            if constexpr (tu_has(K_TNCG)) return launch_bin<NC, K_TNCG, SL, NW>(stream, a, lds, grid);
            else return 1;
    }
}

// ---- register-tile engine (reg_eval.hpp) ---------------------------------------------------------------------------
// A factor row is held by REG_G lanes (NS = 1 or 2 slots per lane); a step covers 64 / REG_G nonzeros.  Tile steps S
// with an instantiated kernel; a bin takes the smallest S that covers its longest row.
#ifndef PMF_REG_G
#define PMF_REG_G 16
#endif
constexpr int REG_G = PMF_REG_G;
constexpr int REG_JG = WAVE / REG_G;
// tile sizes in nonzeros with an instantiated kernel (S = nonzeros / REG_JG steps)
#define PMF_REG_SIZES(X) X(16) X(32) X(48) X(64) X(80) X(96) X(112) X(128) X(144) X(160)
constexpr int REG_NNZ_MAX = 160;
int reg_steps_for(unsigned max_nnz)
{
#define X(NZ) if ((unsigned)(NZ) >= max_nnz) return (NZ) / REG_JG;
    PMF_REG_SIZES(X)
#undef X
    return 0;
}

template <int METHOD, int S, int NS> int launch_reg(hipStream_t stream, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    auto kern = half_sweep_reg_kernel<real_t, METHOD, S, REG_G, NS>;
    static int occ = 0;  // waves per CU the register budget of this instance allows
    if (occ == 0) {
        int n = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), WAVE, 0));
        occ = std::max(1, n);
    }
    const unsigned grid = (unsigned)std::min<size_t>(a.nrows, (size_t)NUM_CU * (size_t)occ * grid_mult);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// longest row (nonzeros) each solver runs from a register tile; beyond it the LDS engine has more waves per CU
#ifndef PMF_REG_MAX_CG
#define PMF_REG_MAX_CG 160
#endif
#ifndef PMF_REG_MAX_TNCG
#define PMF_REG_MAX_TNCG 160
#endif
constexpr int REG_NNZ_MAX_CG = PMF_REG_MAX_CG, REG_NNZ_MAX_TNCG = PMF_REG_MAX_TNCG;
unsigned reg_nnz_max(int method) { return method == POISMF_PG ? REG_NNZ_MAX : method == POISMF_CG ? REG_NNZ_MAX_CG : REG_NNZ_MAX_TNCG; }

template <int S, int NS> int launch_reg_method(hipStream_t stream, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    switch (method) {
        case POISMF_PG:
            if constexpr (tu_has(K_PG)) return launch_reg<K_PG, S, NS>(stream, a, grid_mult);
            else return 1;
        case POISMF_CG:
            if constexpr (tu_has(K_CG) && S * REG_JG <= REG_NNZ_MAX_CG) return launch_reg<K_CG, S, NS>(stream, a, grid_mult);
            else return 1;
        default:
            if constexpr (tu_has(K_TNCG) && S * REG_JG <= REG_NNZ_MAX_TNCG) return launch_reg<K_TNCG, S, NS>(stream, a, grid_mult);
            else return 1;
    }
}

// Several waves per row: the longest share of a row one wave keeps in registers, per solver (CG / TNCG carry more state)
constexpr int REGW_WAVE_NNZ_MAX_PG = 160, REGW_WAVE_NNZ_MAX_CG = 128, REGW_WAVE_NNZ_MAX_TNCG = 96;
unsigned regw_wave_nnz_max(int method)
{
    return (unsigned)(method == POISMF_PG ? REGW_WAVE_NNZ_MAX_PG : method == POISMF_CG ? REGW_WAVE_NNZ_MAX_CG : REGW_WAVE_NNZ_MAX_TNCG);
}
unsigned regw_nnz_max(int method) { return (unsigned)REG_NW_MAX * regw_wave_nnz_max(method); }
// the fewest waves (2, 4, 8) whose shares of a row of max_nnz nonzeros fit
int regw_waves_for(unsigned max_nnz, int method)
{
    for (int nw : { 2, 4, 8 })
        if (max_nnz <= (unsigned)nw * regw_wave_nnz_max(method)) return nw;
    return 0;
}
// tile steps for a row of max_nnz nonzeros split over nw waves (each wave's share is rounded up to whole steps)
int regw_steps_for(unsigned max_nnz, int nw)
{
    const unsigned share = ((max_nnz + (unsigned)nw - 1) / (unsigned)nw + REG_JG - 1) / REG_JG * REG_JG;
    return reg_steps_for(std::max(32u, share));
}

template <int METHOD, int S, int NS, int NW> int launch_regw(hipStream_t stream, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    auto kern = half_sweep_regw_kernel<real_t, METHOD, S, REG_G, NS, NW>;
    static int occ = 0;  // workgroups per CU
    if (occ == 0) {
        int n = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), WAVE * NW, 0));
        occ = std::max(1, n);
    }
    const unsigned grid = (unsigned)std::min<size_t>(a.nrows, (size_t)NUM_CU * (size_t)occ * grid_mult);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * NW), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int S, int NS, int NW> int launch_regw_method(hipStream_t stream, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    if constexpr (S * REG_JG < 32) return 1;
    else switch (method) {
        case POISMF_PG:
            if constexpr (tu_has(K_PG)) return launch_regw<K_PG, S, NS, NW>(stream, a, grid_mult);
            else return 1;
        case POISMF_CG:
            if constexpr (tu_has(K_CG) && S * REG_JG <= REGW_WAVE_NNZ_MAX_CG) return launch_regw<K_CG, S, NS, NW>(stream, a, grid_mult);
            else return 1;
        default:
            if constexpr (tu_has(K_TNCG) && S * REG_JG <= REGW_WAVE_NNZ_MAX_TNCG) return launch_regw<K_TNCG, S, NS, NW>(stream, a, grid_mult);
            else return 1;
    }
}

template <int NS, int NW> int launch_regw_steps_nw(hipStream_t stream, int S, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    switch (S) {
#define X(NZ) case (NZ) / REG_JG: return launch_regw_method<(NZ) / REG_JG, NS, NW>(stream, method, a, grid_mult);
        PMF_REG_SIZES(X)
#undef X
    }
    return 1;
}
template <int NS> int launch_regw_steps(hipStream_t stream, int nw, int S, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    switch (nw) {
        case 2: return launch_regw_steps_nw<NS, 2>(stream, S, method, a, grid_mult);
        case 4: return launch_regw_steps_nw<NS, 4>(stream, S, method, a, grid_mult);
        case 8: return launch_regw_steps_nw<NS, 8>(stream, S, method, a, grid_mult);
    }
    return 1;
}

template <int NS> int launch_reg_steps(hipStream_t stream, int S, int method, const HalfArgs<real_t>& a, unsigned grid_mult)
{
    switch (S) {
#define X(NZ) case (NZ) / REG_JG: return launch_reg_method<(NZ) / REG_JG, NS>(stream, method, a, grid_mult);
        PMF_REG_SIZES(X)
#undef X
    }
    return 1;
}

// Long-row path: rows above this many nonzeros get a whole workgroup of LONG_NW waves (row_eval.hpp, NW > 1).
constexpr unsigned LONG_ROW_NNZ = 8192;
constexpr int LONG_NW = 8;
constexpr int SLOT_ELEMS = (int)(16 / sizeof(real_t));   // elements per 16-byte slot


}  // namespace

// One row-bin launch: everything the planner decided, minus the solver (which selects the translation unit).
struct OneLaunch {
    int reg_S, nw, s_load, spl;   // register-engine steps (0: LDS engine), waves per row, slots per factor row, slots per lane
    bool generic_only;
    hipStream_t main_stream, bin_stream, long_stream;
    size_t lds;
    unsigned grid, grid_mult;
};

#if PMF_TU_KERNELS
namespace {
int launch_one_here(int method, const OneLaunch& o, const HalfArgs<real_t>& a)
{
    int rc = 1;
    if (o.reg_S > 0) {
        if (o.nw > 1) {
            if constexpr (REG_G == 16) rc = launch_regw_steps<1>(o.main_stream, o.nw, o.reg_S, method, a, o.grid_mult);
            else rc = o.s_load <= REG_G ? launch_regw_steps<1>(o.main_stream, o.nw, o.reg_S, method, a, o.grid_mult)
                                        : launch_regw_steps<2>(o.main_stream, o.nw, o.reg_S, method, a, o.grid_mult);
        } else if constexpr (REG_G == 16) rc = launch_reg_steps<1>(o.bin_stream, o.reg_S, method, a, o.grid_mult);
        else rc = o.s_load <= REG_G ? launch_reg_steps<1>(o.bin_stream, o.reg_S, method, a, o.grid_mult)
                                    : launch_reg_steps<2>(o.bin_stream, o.reg_S, method, a, o.grid_mult);
        return rc;
    }
    if (o.nw > 1) {
        switch (o.spl) {
            case 1: rc = launch_method<1 * SLOT_ELEMS, 0, LONG_NW>(o.long_stream, method, a, o.lds, o.grid); break;
            case 2: rc = launch_method<2 * SLOT_ELEMS, 0, LONG_NW>(o.long_stream, method, a, o.lds, o.grid); break;
        }
        return rc;
    }
    if (!o.generic_only && o.s_load == SPECIAL_SL_A) rc = launch_method<SLOT_ELEMS, SPECIAL_SL_A>(o.bin_stream, method, a, o.lds, o.grid);
    else if (!o.generic_only && o.s_load == SPECIAL_SL_B) rc = launch_method<SLOT_ELEMS, SPECIAL_SL_B>(o.bin_stream, method, a, o.lds, o.grid);
    else switch (o.spl) {
        case 1: rc = launch_method<1 * SLOT_ELEMS, 0>(o.bin_stream, method, a, o.lds, o.grid); break;
        case 2: rc = launch_method<2 * SLOT_ELEMS, 0>(o.bin_stream, method, a, o.lds, o.grid); break;
    }
    return rc;
}
}  // namespace
#endif
#if PMF_TU == 1
int pmf_launch_one_tu1(int method, const OneLaunch& o, const HalfArgs<real_t>& a) { return launch_one_here(method, o, a); }
#elif PMF_TU == 2
int pmf_launch_one_tu2(int method, const OneLaunch& o, const HalfArgs<real_t>& a) { return launch_one_here(method, o, a); }
#elif PMF_TU == 3
int pmf_launch_one_tu3(int method, const OneLaunch& o, const HalfArgs<real_t>& a) { return launch_one_here(method, o, a); }
#endif
#if PMF_TU == 0
int pmf_launch_one_tu1(int method, const OneLaunch& o, const HalfArgs<real_t>& a);
int pmf_launch_one_tu2(int method, const OneLaunch& o, const HalfArgs<real_t>& a);
int pmf_launch_one_tu3(int method, const OneLaunch& o, const HalfArgs<real_t>& a);
static int launch_one(int method, const OneLaunch& o, const HalfArgs<real_t>& a)
{
    return method == POISMF_PG ? pmf_launch_one_tu3(method, o, a) : method == POISMF_CG ? pmf_launch_one_tu2(method, o, a) : pmf_launch_one_tu1(method, o, a);
}
#elif PMF_TU == -1
static int launch_one(int method, const OneLaunch& o, const HalfArgs<real_t>& a) { return launch_one_here(method, o, a); }
#endif

namespace {

// column-sum kernels: elements per lane in the plain lane <-> element layout
int nc_for_k(size_t k) { return k <= 64 ? 1 : (k <= 128 ? 2 : (k <= 256 ? 4 : (k <= 512 ? 8 : 0))); }
// row kernels: 16-byte slots per lane (slot layout of row_eval.hpp); 0 = unsupported
int slots_per_lane(size_t k)
{
    const size_t s_load = (k * sizeof(real_t) + 15) / 16;
    return s_load <= 64 ? 1 : (s_load <= 128 ? 2 : 0);
}

template <int NC> int launch_colsum(poismf_hip_session* s, const real_t* M, size_t n, real_t l1, real_t scale, int nscale)
{
    const int nw = (int)std::min<size_t>((size_t)s->colsum_waves, std::max<size_t>((n + COLSUM_BLOCK_WAVES - 1) / COLSUM_BLOCK_WAVES, 1));
    hipLaunchKernelGGL((colsum_partial_kernel<real_t, NC>), dim3(nw), dim3(WAVE * COLSUM_BLOCK_WAVES), 0, s->stream, M, n, (int)s->k,
                       s->d_partial);
    hipLaunchKernelGGL((colsum_final_kernel<real_t, NC>), dim3(1), dim3(WAVE * COLSUM_FINAL_WAVES), 0, s->stream, s->d_partial, nw, (int)s->k, l1,
                       scale, nscale, s->d_bsum);
    HIP_TRY(hipGetLastError());
    return 0;
}

int colsum(poismf_hip_session* s, const real_t* M, size_t n, real_t l1, real_t scale, int nscale)
{
    switch (nc_for_k(s->k)) {
        case 1: return launch_colsum<1>(s, M, n, l1, scale, nscale);
        case 2: return launch_colsum<2>(s, M, n, l1, scale, nscale);
        case 4: return launch_colsum<4>(s, M, n, l1, scale, nscale);
        case 8: return launch_colsum<8>(s, M, n, l1, scale, nscale);
    }
    return 1;
}

}  // namespace

#if PMF_TU_HOST
extern "C" {

int poismf_hip_session_create(poismf_hip_session** out, int device, void* stream, const real_t* Xr,
                              const sparse_ix* Xr_indptr, const sparse_ix* Xr_indices, const real_t* Xc,
                              const sparse_ix* Xc_indptr, const sparse_ix* Xc_indices, size_t dimA, size_t dimB, size_t k,
                              size_t rowA_begin, size_t rowA_end, size_t rowB_begin, size_t rowB_end)
{
    *out = nullptr;
    if (k == 0 || slots_per_lane(k) == 0 || nc_for_k(k) == 0) {
        fprintf(stderr, "poismf_hip: k = %zu is outside the supported range (1..%d)\n", k, 128 * SLOT_ELEMS);
        return 1;
    }
    if (rowA_end > dimA || rowB_end > dimB || rowA_begin > rowA_end || rowB_begin > rowB_end) return 1;
    HIP_TRY(hipSetDevice(device));
    poismf_hip_session* s = new (std::nothrow) poismf_hip_session();
    if (!s) return 1;
    s->device = device;
    s->stream = (hipStream_t)stream;
    if (s->stream == nullptr) {  // no stream given: own one, so that fork / join with the auxiliary stream is explicit
        if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return 1; }
        s->owns_stream = true;
    }
    s->dimA = dimA; s->dimB = dimB; s->k = k;
    auto fail = [&]() { poismf_hip_session_destroy(s); return 1; };
    // behind each factor: one all-zero row (the register engine points the unused steps of a row at it) + 16 B so that
    // the last 16-byte slot of the last row stays in bounds
    const size_t slack = k * sizeof(real_t) + 16;
    if (hipMalloc(&s->dA, dimA * k * sizeof(real_t) + slack) != hipSuccess) return fail();
    if (hipMalloc(&s->dB, dimB * k * sizeof(real_t) + slack) != hipSuccess) return fail();
    if (hipMemsetAsync(s->dA, 0, dimA * k * sizeof(real_t) + slack, s->stream) != hipSuccess) return fail();
    if (hipMemsetAsync(s->dB, 0, dimB * k * sizeof(real_t) + slack, s->stream) != hipSuccess) return fail();
    {
        // A gathered row of B bytes at an arbitrary 8-byte offset touches (B + 120) / 128 lines of 128 bytes on average;
        // in a copy whose rows start on line boundaries it touches ceil(B / 128).  k = 50 fp32: 2.5 -> 2 lines.
        const size_t rowb = k * sizeof(real_t);
        size_t padb = (rowb + 127) / 128 * 128;
        static const bool no_pad = getenv("POISMF_HIP_NO_PAD") != nullptr;  // testing knob
        const bool line_pad = !no_pad && padb != rowb && (double)padb <= 0.9 * (double)(rowb + 120);
        // and a row that does not end on a 16-byte slot boundary is padded to one in any case: the gathers fetch whole
        // slots and rely on the excess of the last one being zero
        if (!line_pad) padb = (rowb + 15) / 16 * 16;
        if (padb != rowb) {
            s->ld = padb / sizeof(real_t);
            const size_t pslack = padb + 16;
            if (hipMalloc(&s->dAp, dimA * padb + pslack) != hipSuccess) return fail();
            if (hipMalloc(&s->dBp, dimB * padb + pslack) != hipSuccess) return fail();
            if (hipMemsetAsync(s->dAp, 0, dimA * padb + pslack, s->stream) != hipSuccess) return fail();
            if (hipMemsetAsync(s->dBp, 0, dimB * padb + pslack, s->stream) != hipSuccess) return fail();
        }
    }
    if (hipMalloc(&s->d_bsum, k * sizeof(real_t) + slack) != hipSuccess) return fail();
    if (hipMalloc(&s->d_partial, (size_t)s->colsum_waves * k * sizeof(real_t)) != hipSuccess) return fail();
    if (hipMalloc(&s->d_counter, sizeof(unsigned)) != hipSuccess) return fail();
    if (hipMalloc(&s->d_queue, sizeof(unsigned) * MAX_LAUNCHES) != hipSuccess) return fail();
    if (hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking) != hipSuccess) return fail();
    if (hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess) return fail();
    if (hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess) return fail();
    // half 0 updates B: rows of the CSC; half 1 updates A: rows of the CSR
    if (Xc_indptr != nullptr &&
        build_half(s->half[0], s->stream, Xc, Xc_indptr, Xc_indices, dimB, dimA, rowB_begin, rowB_end)) return fail();
    if (build_half(s->half[1], s->stream, Xr, Xr_indptr, Xr_indices, dimA, dimB, rowA_begin, rowA_end)) return fail();
    *out = s;
    return 0;
}

void poismf_hip_session_destroy(poismf_hip_session* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->stream);
    if (s->aux_stream) (void)hipStreamSynchronize(s->aux_stream);
    for (auto& p : s->prof) { (void)hipEventDestroy(p.t0); (void)hipEventDestroy(p.t1); }
    s->prof.clear();
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->aux_stream) (void)hipStreamDestroy(s->aux_stream);
    if (s->owns_stream) (void)hipStreamDestroy(s->stream);
    free_half(s->half[0]);
    free_half(s->half[1]);
    if (s->dA) (void)hipFree(s->dA);
    if (s->dB) (void)hipFree(s->dB);
    if (s->dAp) (void)hipFree(s->dAp);
    if (s->dBp) (void)hipFree(s->dBp);
    if (s->d_bsum) (void)hipFree(s->d_bsum);
    if (s->d_partial) (void)hipFree(s->d_partial);
    if (s->d_counter) (void)hipFree(s->d_counter);
    if (s->d_queue) (void)hipFree(s->d_queue);
    delete s;
}

// Whoever asks for the device pointers may write through them: the padded gather copies are re-derived afterwards.
real_t* poismf_hip_session_A(poismf_hip_session* s) { s->padded_fresh[1] = false; return s->dA; }
real_t* poismf_hip_session_B(poismf_hip_session* s) { s->padded_fresh[0] = false; return s->dB; }
size_t poismf_hip_session_nnz(poismf_hip_session* s, int which) { return s->half[which ? 1 : 0].nnz; }

#ifdef PMF_TIMING
// development-only export (not in the header): read and reset the phase timers
__attribute__((visibility("default"))) void poismf_hip_debug_timing(unsigned long long* out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pmf_timing), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pmf_timing), z, sizeof(z));
}
#endif

// Largest distance in ulps between this library's double log (wave_ops.hpp) and the device library's over n sample
// arguments, and the number of special arguments (+-0, -1, +-inf, NaN) on which they disagree.
int poismf_hip_selftest_log(size_t n, unsigned long long* worst_ulp, unsigned* mismatched_specials)
{
    unsigned long long* d_w = nullptr;
    unsigned* d_m = nullptr;
    HIP_TRY(hipMalloc(&d_w, sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&d_m, sizeof(unsigned)));
    HIP_TRY(hipMemset(d_w, 0, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d_m, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(selftest_log_kernel, dim3(NUM_CU * 8), dim3(256), 0, 0, (unsigned long long)n, d_w, d_m);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(worst_ulp, d_w, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mismatched_specials, d_m, sizeof(unsigned), hipMemcpyDeviceToHost));
    (void)hipFree(d_w);
    (void)hipFree(d_m);
    return 0;
}

int poismf_hip_session_set_factors(poismf_hip_session* s, const real_t* A_host, const real_t* B_host)
{
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync(s->dA, A_host, s->dimA * s->k * sizeof(real_t), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->dB, B_host, s->dimB * s->k * sizeof(real_t), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->padded_fresh[0] = s->padded_fresh[1] = false;
    return 0;
}

int poismf_hip_session_get_factors(poismf_hip_session* s, real_t* A_host, real_t* B_host)
{
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync(A_host, s->dA, s->dimA * s->k * sizeof(real_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(B_host, s->dB, s->dimB * s->k * sizeof(real_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

void poismf_hip_session_profile(poismf_hip_session* s, int enable)
{
    (void)hipStreamSynchronize(s->stream);
    for (auto& p : s->prof) { (void)hipEventDestroy(p.t0); (void)hipEventDestroy(p.t1); }
    s->prof.clear();
    s->profiling = enable != 0;
    for (Half& h : s->half) {
        const size_t n = h.row_end - h.row_begin;
        if (s->profiling && h.d_eval_rows == nullptr && n > 0 && hipMalloc(&h.d_eval_rows, sizeof(unsigned) * n) != hipSuccess) h.d_eval_rows = nullptr;
        if (h.d_eval_rows != nullptr) (void)hipMemsetAsync(h.d_eval_rows, 0, sizeof(unsigned) * n, s->stream);
    }
}

int poismf_hip_session_kernel_time(poismf_hip_session* s, int which, double* total_ms, size_t* launches)
{
    HIP_TRY(hipStreamSynchronize(s->stream));
    double tot = 0;
    size_t n = 0;
    for (auto& p : s->prof) {
        if (p.which != which) continue;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.t0, p.t1));
        tot += ms;
        n++;
    }
    *total_ms = tot;
    *launches = n;
    return 0;
}

int poismf_hip_session_eval_stats(poismf_hip_session* s, int which, unsigned long long* tile_passes, unsigned long long* nnz_passes)
{
    HIP_TRY(hipStreamSynchronize(s->stream));
    Half& h = s->half[which ? 1 : 0];
    const size_t n = h.row_end - h.row_begin;
    *tile_passes = 0;
    *nnz_passes = 0;
    if (h.d_eval_rows == nullptr || n == 0) return 0;
    std::vector<unsigned> ev(n);
    HIP_TRY(hipMemcpy(ev.data(), h.d_eval_rows, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        *tile_passes += ev[i];
        *nnz_passes += (unsigned long long)ev[i] * h.row_nnz[i];
    }
    return 0;
}

// bsum_override != nullptr: use this HOST k-vector (already carrying l1 and any PG scaling) instead of the column
// sums of the fixed factor; neg_step_override then replaces -step_size as the PG scale of the per-row Bsum_w.
static int half_sweep_impl(poismf_hip_session* s, int which, const poismf_hip_params* p, real_t step_size, real_t cnst_div,
                           size_t* n_unchanged, const real_t* bsum_override, real_t neg_step_override)
{
    HIP_TRY(hipSetDevice(s->device));
    which = which ? 1 : 0;
    Half& h = s->half[which];
    real_t* M = which ? s->dA : s->dB;
    const real_t* F = which ? s->dB : s->dA;
    const size_t dimF = which ? s->dimB : s->dimA;
    const bool is_pg = p->method == POISMF_PG;
    const bool weighted = p->w_mult != (real_t)1.;

    // column sums of the fixed factor (+ l1), with the PG pre-scaling when w == 1:
    //   B half: * (-step)            ref: src/poismf.c:523-524
    //   A half: * (-step) twice      ref: src/poismf.c:573-577 (quirk Q1)
    real_t neg_step = -step_size;
    if (bsum_override != nullptr) {
        neg_step = neg_step_override;
        HIP_TRY(hipMemcpyAsync(s->d_bsum, bsum_override, s->k * sizeof(real_t), hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));  // the caller's host vector may change right after this call
    } else {
        int nscale = 0;
        if (is_pg && !weighted) nscale = which ? 2 : 1;
        if (colsum(s, F, dimF, p->l1_reg, neg_step, nscale)) return 1;
    }

    // the gathers read the line-padded copy of the fixed factor when the session keeps one
    const real_t* Fg = F;
    size_t ldF = s->k;
    real_t* Mp = nullptr;
    if (s->ld != 0) {
        // The padded copy of F is current only if this session's own previous half-sweep rewrote ALL of F (its row
        // kernels store every updated row to both copies).  Anything else -- factors set by the caller, a shard
        // exchange between GPUs writing into the compact factor -- is picked up by re-padding the whole factor.
        real_t* Fp = which ? s->dBp : s->dAp;
        if (!s->padded_fresh[which ? 0 : 1]) {
            const size_t total = dimF * s->k;
            const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, (size_t)NUM_CU * 16);
            if (blocks > 0) hipLaunchKernelGGL((repad_kernel<real_t>), dim3(blocks), dim3(256), 0, s->stream, F, Fp, dimF, (int)s->k, (int)s->ld);
            HIP_TRY(hipGetLastError());
        }
        Fg = Fp;
        ldF = s->ld;
        Mp = which ? s->dAp : s->dBp;
        s->padded_fresh[which ? 1 : 0] = h.row_begin == 0 && h.row_end == h.dimM;
    }

    HalfArgs<real_t> a;
    a.M = M; a.F = Fg;
    a.Mp = Mp; a.ldM = (int)s->ld;
    a.indptr = h.d_indptr; a.indices = h.d_indices; a.values = h.d_values; a.perm = h.d_perm; a.desc = h.d_desc;
    a.row_offset = (unsigned)h.row_begin;
    a.bsum = s->d_bsum;
    a.P.l2 = p->l2_reg; a.P.w = p->w_mult;
    a.P.step = step_size * p->w_mult;  // ref: src/poismf.c:151
    a.P.cnst_div = cnst_div;
    a.P.neg_step = neg_step;
    a.P.maxupd = (int)std::min<size_t>(p->maxupd, 0x7fffffff);
    a.P.limit_step = p->limit_step;
    a.P.max_cg_it = (int)std::max(1.0, std::min(50.0, (double)(real_t)s->k / 2.0));  // ref: src/poismf.c:342
    a.reuse_prev = p->reuse_prev;
    a.early_stop = (p->method == POISMF_TNCG) && p->early_stop && n_unchanged != nullptr;
    a.n_unchanged = s->d_counter;
    a.eval_rows = s->profiling ? h.d_eval_rows : nullptr;
    if (a.early_stop) HIP_TRY(hipMemsetAsync(s->d_counter, 0, sizeof(unsigned), s->stream));

    const bool single_pass = is_pg && p->maxupd <= 1 && !weighted;
    ProfRec rec{};
    if (s->profiling) {
        HIP_TRY(hipEventCreate(&rec.t0));
        HIP_TRY(hipEventCreate(&rec.t1));
        rec.which = which;
        HIP_TRY(hipEventRecord(rec.t0, s->stream));
    }
    // Consecutive bins that end up with the same tile geometry (all streamed bins; every bin of a single-pass
    // solver) are merged into one launch.
    struct Launch { unsigned begin, count; TileGeom geom; int nw; int reg_S; };
    std::vector<Launch> launches;
    static const bool no_reg = getenv("POISMF_HIP_NO_REGTILE") != nullptr;  // testing knob: LDS engine for every row
    // register engine: factor rows of at most 16 slots, and 24-bit row ids / 32-bit byte offsets into the factor
    const bool reg_ok = !no_reg && (s->k * sizeof(real_t) + 15) / 16 <= 16 &&
                        dimF < ((size_t)1 << 24) && (dimF + 1) * ldF * sizeof(real_t) + 16 < ((size_t)1 << 32);
    static const bool no_long = getenv("POISMF_HIP_NO_LONGROW") != nullptr;  // testing knob
    unsigned long_thr = LONG_ROW_NNZ;
    if (const char* e = getenv("POISMF_HIP_LONGROW_NNZ")) long_thr = (unsigned)std::max(64, atoi(e));  // testing knob
    for (const Bin& b : h.bins) {
        TileGeom g = plan_geom(s->k, b.cls, single_pass, p->method == POISMF_CG && p->limit_step);
        if (single_pass) { g.resident = 0; g.prefetch = prefetch_enabled() ? 1 : 0; }  // one pass: "gather once" and "stream" are the same thing
        if (reg_ok && b.cls <= reg_nnz_max(p->method)) {
            // short rows: the tile lives in registers (reg_eval.hpp); bins sharing a step count share a launch
            // (a bin of a few thousand rows is not worth a launch of its own: it rides along with the next longer size)
            const int S = reg_steps_for(b.max_nnz);
            if (!launches.empty() && launches.back().nw == 1 && launches.back().reg_S >= S &&
                (launches.back().reg_S == S || b.count < 4096u) && launches.back().begin + launches.back().count == b.begin)
                launches.back().count += b.count;
            else
                launches.push_back({ b.begin, b.count, g, 1, S });
            continue;
        }
        if (reg_ok && b.cls <= regw_nnz_max(p->method)) {
            // medium rows: 2, 4 or 8 waves share a row, each keeps its part of the tile in registers
            const int nw = regw_waves_for(b.cls, p->method);
            const int S = regw_steps_for(b.max_nnz, nw);
            if (!launches.empty() && launches.back().nw == nw && launches.back().reg_S >= S &&
                (launches.back().reg_S == S || b.count < 2048u) && launches.back().begin + launches.back().count == b.begin)
                launches.back().count += b.count;
            else
                launches.push_back({ b.begin, b.count, g, nw, S });
            continue;
        }
        if (!no_long && b.cls > long_thr) {
            // a workgroup of LONG_NW waves per row; every wave streams its own chunks: size the chunk so that
            // LONG_NW private tiles and the reduction scratch fit in one CU's LDS
            g.resident = 0;
            g.prefetch = prefetch_enabled() ? 1 : 0;
            g.pq_cap = 0;
            int cap = 128;
            for (;;) {
                g.cap = cap;
                if (cap <= 16 || lds_bytes_per_block(g, sizeof(real_t), LONG_NW) <= 150 * 1024) break;
                cap -= 16;
            }
            if (!launches.empty() && launches.back().reg_S == 0 && launches.back().nw == LONG_NW && launches.back().begin + launches.back().count == b.begin)
                launches.back().count += b.count;
            else
                launches.push_back({ b.begin, b.count, g, LONG_NW, 0 });
            continue;
        }
        if (!launches.empty() && launches.back().reg_S == 0 && launches.back().geom.cap == g.cap &&
            launches.back().geom.resident == g.resident && (g.resident == 0) && g.pq_cap == 0 && launches.back().geom.pq_cap == 0 && launches.back().nw == 1 &&
            launches.back().begin + launches.back().count == b.begin)
            launches.back().count += b.count;
        else
            launches.push_back({ b.begin, b.count, g, 1, 0 });
    }
    static const bool static_rows = getenv("POISMF_HIP_STATIC_ROWS") != nullptr;  // testing knob
    const bool dynamic = !is_pg && !static_rows && launches.size() <= (size_t)MAX_LAUNCHES;
    if (dynamic) HIP_TRY(hipMemsetAsync(s->d_queue, 0, sizeof(unsigned) * MAX_LAUNCHES, s->stream));
    // the few workgroup-per-row launches of the power-law tail occupy a few dozen CUs for a long time: run them on a
    // second stream beside the other bins (fork after the column sums, join before anything reads the result)
    // Several launches per half: they also alternate between the two streams (each to the one with less work queued so
    // far), so that the tail of one bin overlaps the start of the next.
    // (off by default: overlapping launches make the per-kernel durations of a profile overlap too)
    static const bool no_fork = getenv("POISMF_HIP_NO_FORK") != nullptr;  // testing knob
    static const bool fork_bins = getenv("POISMF_HIP_FORK_BINS") != nullptr;  // tuning knob
    bool any_long = false;
    for (const Launch& L : launches) any_long = any_long || (L.nw > 1 && L.reg_S == 0);
    const bool forked = !no_fork && launches.size() > 1 && (any_long || fork_bins);
    hipStream_t long_stream = forked ? s->aux_stream : s->stream;
    double queued[2] = { 0.0, 0.0 };
    if (forked) {
        HIP_TRY(hipEventRecord(s->ev_fork, s->stream));
        HIP_TRY(hipStreamWaitEvent(s->aux_stream, s->ev_fork, 0));
    }
    int launch_no = 0;
    for (const Launch& L : launches) {
        a.queue = dynamic ? s->d_queue + launch_no : nullptr;
        launch_no++;
        a.perm_begin = L.begin;
        a.nrows = L.count;
        a.geom = L.geom;
        a.geom.zero_row = (unsigned)dimF;
        a.geom.ldF = (int)ldF;
        const size_t lds = lds_bytes_per_block(a.geom, sizeof(real_t), L.nw);
        const unsigned waves_per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(16, LDS_PER_CU / lds));
        // Waves launched per resident wave slot.  Rows pulled from the queue balance themselves: 2 is enough.  Rows dealt
        // out statically come in nnz-descending order, so wave 0 always gets the longest of each round; many short
        // waves let the dispatcher even that out (measured on C2, PG(10): 2 -> 1.214 ms, 8 -> 1.165, 32 -> 1.146).
        // The single-wave register kernels are always dealt out this way: with ~1 row per wave the hardware dispatcher IS
        // the queue (CG fp32 on C2: 3.87 ms with tickets, 3.35 ms without).
        const bool one_wave_reg = L.reg_S > 0 && L.nw == 1;
        if (one_wave_reg) a.queue = nullptr;
        unsigned grid_mult = one_wave_reg ? 32 : 2;
        if (const char* e = getenv("POISMF_HIP_GRID_MULT")) grid_mult = (unsigned)std::max(1, atoi(e));  // tuning knob
        const unsigned grid = (unsigned)std::min<size_t>(L.count, (size_t)NUM_CU * waves_per_cu * grid_mult);
        int rc = 1;
        const int lane_stream = (forked && fork_bins && L.nw == 1 && queued[1] < queued[0]) ? 1 : 0;
        hipStream_t bin_stream = lane_stream ? s->aux_stream : s->stream;
        queued[L.nw > 1 ? 1 : lane_stream] += (double)L.count * (double)std::max(16, L.reg_S > 0 ? L.reg_S * REG_JG : L.geom.cap);
        {
            OneLaunch o;
            o.reg_S = L.reg_S; o.nw = L.nw; o.s_load = a.geom.s_load; o.spl = slots_per_lane(s->k);
            static const bool generic_only = getenv("POISMF_HIP_GENERIC") != nullptr;  // testing knob: skip the specialisations
            o.generic_only = generic_only;
            o.main_stream = s->stream; o.bin_stream = bin_stream; o.long_stream = long_stream;
            o.lds = lds; o.grid = grid; o.grid_mult = grid_mult;
            rc = launch_one(p->method, o, a);
        }
        if (rc) return 1;
    }
    if (forked) {
        HIP_TRY(hipEventRecord(s->ev_join, s->aux_stream));
        HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_join, 0));
    }
    if (s->profiling) {
        HIP_TRY(hipEventRecord(rec.t1, s->stream));
        s->prof.push_back(rec);
    }
    if (a.early_stop) {
        unsigned cnt = 0;
        HIP_TRY(hipMemcpyAsync(&cnt, s->d_counter, sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        *n_unchanged = cnt;
    }
    return 0;
}

int poismf_hip_half_sweep(poismf_hip_session* s, int which, const poismf_hip_params* p, real_t step_size, real_t cnst_div,
                          size_t* n_unchanged)
{
    return half_sweep_impl(s, which, p, step_size, cnst_div, n_unchanged, nullptr, (real_t)0);
}

// -------------------------------------------------------------------------------------------------
// run_poismf: the drop-in                                   ref: src/poismf.c:435-632
// -------------------------------------------------------------------------------------------------
static volatile sig_atomic_t g_should_stop = 0;
static bool g_handle_locked = false;
static std::mutex g_handle_mutex;
static void on_sigint(int)
{
    g_should_stop = 1;  // the reference also prints here; fprintf is not async-signal-safe, so run_poismf reports it
}

int run_poismf(real_t* A, real_t* Xr, sparse_ix* Xr_indptr, sparse_ix* Xr_indices, real_t* B, real_t* Xc,
               sparse_ix* Xc_indptr, sparse_ix* Xc_indices, const size_t dimA, const size_t dimB, const size_t k,
               const real_t l2_reg, const real_t l1_reg, const real_t w_mult, real_t step_size, const int method,
               const bool limit_step, const size_t numiter, const size_t maxupd, const bool early_stop,
               const bool reuse_prev, const bool handle_interrupt, const int nthreads)
{
    (void)nthreads;
    typedef void (*sig_fn)(int);
    sig_fn old_handler = nullptr;
    bool has_lock = false;
    {
        std::lock_guard<std::mutex> lk(g_handle_mutex);  // ref: :446-455
        if (!g_handle_locked) {
            g_handle_locked = true;
            has_lock = true;
            g_should_stop = 0;
            old_handler = signal(SIGINT, on_sigint);
        }
    }

    int ret_code = 0;
    poismf_hip_session* s = nullptr;
    int device = 0;
    if (const char* e = getenv("POISMF_HIP_DEVICE")) device = atoi(e);

    if (poismf_hip_session_create(&s, device, nullptr, Xr, Xr_indptr, Xr_indices, Xc, Xc_indptr, Xc_indices, dimA, dimB, k, 0,
                                  dimA, 0, dimB) ||
        poismf_hip_session_set_factors(s, A, B)) {
        fprintf(stderr, "Error: out of memory.\n");  // ref: :501
        ret_code = 1;
    } else {
        poismf_hip_params p;
        p.l2_reg = l2_reg; p.l1_reg = l1_reg; p.w_mult = w_mult; p.step_size = step_size;
        p.method = method; p.limit_step = limit_step; p.maxupd = maxupd;
        p.early_stop = early_stop; p.reuse_prev = reuse_prev;
        const bool tn_stop = (method == POISMF_TNCG) && early_stop;
        bool stopped_earlyA = false, stopped_earlyB = false;
        bool failed = false;

        for (size_t it = 0; it < numiter && !failed; it++) {
            if (g_should_stop) break;
            // quirk Q6: the divisor uses the step before halving and is reused by the A half
            const real_t cnst_div = 1. / (1. + 2. * l2_reg * step_size);

            // ---- B half first (quirk Q5) ----
            if (!(method == POISMF_TNCG && stopped_earlyB)) {
                size_t unchanged = 0;
                if (poismf_hip_half_sweep(s, 0, &p, step_size, cnst_div, tn_stop ? &unchanged : nullptr)) { failed = true; break; }
                if (tn_stop) stopped_earlyB = ((double)unchanged / (double)dimB) >= .95;  // ref: :401-403 (quirk Q7)
            }
            if (method == POISMF_PG) step_size *= 0.5;  // ref: :532-533
            if (hipStreamSynchronize(s->stream) != hipSuccess) { failed = true; break; }
            if (g_should_stop) break;

            // ---- A half ----
            if (!(method == POISMF_TNCG && stopped_earlyA)) {
                size_t unchanged = 0;
                if (poismf_hip_half_sweep(s, 1, &p, step_size, cnst_div, tn_stop ? &unchanged : nullptr)) { failed = true; break; }
                if (tn_stop) stopped_earlyA = ((double)unchanged / (double)dimA) >= .95;
            }
            if (hipStreamSynchronize(s->stream) != hipSuccess) { failed = true; break; }
            if (stopped_earlyA && stopped_earlyB) break;
        }
        if (failed || poismf_hip_session_get_factors(s, A, B)) {
            fprintf(stderr, "Error: out of memory.\n");
            ret_code = 1;
        }
    }
    poismf_hip_session_destroy(s);

    {
        std::lock_guard<std::mutex> lk(g_handle_mutex);  // ref: :618-630
        const bool stopped = g_should_stop != 0;
        if (stopped) fprintf(stderr, "Error: procedure was interrupted\n");
        if (stopped && ret_code != 1) ret_code = 2;
        if (has_lock) {
            signal(SIGINT, old_handler);
            g_handle_locked = false;
            g_should_stop = 0;
        }
        if (stopped && !handle_interrupt) raise(SIGINT);
    }
    return ret_code;
}

// -------------------------------------------------------------------------------------------------
// factors_multiple: latent factors of new rows with B fixed          ref: src/pred.c:66-199
// -------------------------------------------------------------------------------------------------
int factors_multiple(real_t* A, real_t* B, real_t* Bsum, real_t* Amean, real_t* Xr, sparse_ix* Xr_indptr,
                     sparse_ix* Xr_indices, int k, size_t dimA, real_t l2_reg, real_t w_mult, real_t step_size,
                     size_t niter, size_t maxupd, int method, bool limit_step, bool reuse_mean, int nthreads)
{
    (void)nthreads;
    const size_t ks = (size_t)k;
    const size_t nnz = Xr_indptr[dimA];
    // rows start at the mean of the fitted A, except TNCG without reuse_mean (1e-3, set in the kernel); ref: :144-147
    if (reuse_mean || method != POISMF_TNCG)
        for (size_t r = 0; r < dimA; r++) memcpy(A + r * ks, Amean, ks * sizeof(real_t));
    if (nnz == 0) {  // every row is empty: all three drivers zero such rows (quirk Q7)
        memset(A, 0, dimA * ks * sizeof(real_t));
        return 0;
    }
    size_t dimB = 0;  // the reference never needs the number of items; the device copy of B needs the rows in use
    for (size_t i = 0; i < nnz; i++) dimB = std::max(dimB, (size_t)Xr_indices[i] + 1);

    int device = 0;
    if (const char* e = getenv("POISMF_HIP_DEVICE")) device = atoi(e);
    poismf_hip_session* s = nullptr;
    int rc = 0;
    std::vector<real_t> bs(ks);
    if (poismf_hip_session_create(&s, device, nullptr, Xr, Xr_indptr, Xr_indices, nullptr, nullptr, nullptr, dimA, dimB, ks, 0,
                                  dimA, 0, 0) ||
        hipMemcpy(s->dA, A, dimA * ks * sizeof(real_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(s->dB, B, dimB * ks * sizeof(real_t), hipMemcpyHostToDevice) != hipSuccess) {
        rc = 1;
    } else {
        poismf_hip_params p;
        p.l2_reg = l2_reg; p.l1_reg = 0; p.w_mult = w_mult; p.step_size = step_size;
        p.method = method; p.limit_step = limit_step; p.maxupd = maxupd;
        p.early_stop = 0; p.reuse_prev = reuse_mean;
        const bool weighted = w_mult != (real_t)1.;
        if (method == POISMF_PG) {                                    // ref: :152-169
            const real_t step0 = step_size;
            for (size_t it = 0; it < niter && !rc; it++) {
                for (size_t c = 0; c < ks; c++) bs[c] = weighted ? Bsum[c] : Bsum[c] * (-step_size);
                const real_t cnst_div = 1. / (1. + 2. * l2_reg * step_size);
                // w != 1: Bsum_w was scaled by -step at set-up (ref: :121-122) and again by -step here (ref: :162)
                rc = half_sweep_impl(s, 1, &p, step_size, cnst_div, nullptr, bs.data(), (-step0) * (-step_size));
                step_size *= 0.5;
            }
        } else {
            for (size_t c = 0; c < ks; c++) bs[c] = Bsum[c];
            if (method == POISMF_CG) p.maxupd = maxupd * niter;      // ref: :175-178
            rc = half_sweep_impl(s, 1, &p, step_size, (real_t)1, nullptr, bs.data(), -step_size);
        }
        if (!rc && (hipStreamSynchronize(s->stream) != hipSuccess ||
                    hipMemcpy(A, s->dA, dimA * ks * sizeof(real_t), hipMemcpyDeviceToHost) != hipSuccess))
            rc = 1;
    }
    poismf_hip_session_destroy(s);
    if (rc) fprintf(stderr, "Error: out of memory.\n");
    return rc ? 1 : 0;
}

}  // extern "C"
#endif  // PMF_TU_HOST
