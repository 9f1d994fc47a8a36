// poismf_hip_host.hip -- host side of the MI355X implementation of poismf's alternating factor-update path: the outer
// A/B alternation, step schedule, early-stop logic, SIGINT plumbing and return codes of run_poismf
// (ref: src/poismf.c:435-632), the device-resident session used by bench.py and by the one-process-per-GPU driver,
// the planning of a half-sweep into row-bin launches, and the small dense kernels (column sums, padded copies).
// The row kernels are poismf_hip.hip (one translation unit per inner solver); the C-ABI is include/poismf_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <csignal>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "plan.hpp"

// the device-visible twin of the interrupt flag (pinned host memory; see on_sigint below): one word, allocated on first use, never freed
static volatile unsigned* g_stop_word = nullptr;

static int launch_one(int method, const OneLaunch& o, const HalfArgs<real_t>& a)
{
    return method == POISMF_PG ? pmf_launch_one_tu3(method, o, a) : method == POISMF_CG ? pmf_launch_one_tu2(method, o, a) :
           method == POISMF_EVAL ? pmf_launch_one_tu4(method, o, a) : pmf_launch_one_tu1(method, o, a);
}

#define PMF_EW _Pragma("unroll") for (int i = 0; i < NC; i++)

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
__global__ __launch_bounds__(WAVE* COLSUM_BLOCK_WAVES) void colsum_partial_kernel(const T* M, size_t n, int k, T* partial, unsigned block0, unsigned nblocks)
{
    // (block0 / nblocks: this launch computes blocks [block0, block0 + gridDim.x) of the nblocks the whole sum is cut into -- a block's
    // partial sum depends on nblocks and on its own number alone, so ANY subset of blocks, computed anywhere, gives the same bits: what lets
    // the ranks of a multi-GPU run share the first stage, poismf_hip_session_colsum_partial)
    __shared__ T part[COLSUM_BLOCK_WAVES][NC * WAVE];
    const int lane = lane_id();
    const int w = (int)(threadIdx.x / WAVE);
    T acc[NC];
    PMF_EW acc[i] = (T)0;
    // four rows in flight per wave (the loop is latency-bound: one 200-byte row per trip); added in row order as before
    const unsigned bid = blockIdx.x + block0;
    const size_t stride = (size_t)nblocks * COLSUM_BLOCK_WAVES;
    size_t r = (size_t)bid * COLSUM_BLOCK_WAVES + w;
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
                partial[(size_t)bid * k + c] = s;
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
// serve.hip (cores on device-resident factors)
int poismf_hip_serve_predict(const real_t* dA, const real_t* dB, const sparse_ix* ixA, const sparse_ix* ixB, size_t n, int k, real_t* out,
                             size_t* max_a, size_t* max_b);
int poismf_hip_serve_topn(const real_t* d_a, const real_t* dB, int k, const sparse_ix* include_ix, size_t n_include,
                          const sparse_ix* exclude_ix, size_t n_exclude, sparse_ix* outp_ix, real_t* outp_score, size_t n_top, size_t n);
int poismf_hip_serve_topn_check(const sparse_ix*& include_ix, size_t n_include, const sparse_ix*& exclude_ix, size_t n_exclude, size_t n_top, size_t n);
// coo_convert.hip (rocPRIM-based helpers)
int poismf_hip_device_sort_rows(const unsigned long long* d_indptr, size_t nloc, unsigned base, unsigned* d_perm, unsigned* d_len_sorted,
                                hipStream_t stream);
int poismf_hip_device_narrow(const unsigned long long* d_src, size_t n, unsigned* d_dst, hipStream_t stream);
int poismf_hip_device_coo_to_cs(const unsigned* d_major, const unsigned* d_minor, const real_t* d_val, size_t n, size_t major_begin,
                                size_t major_end, unsigned* out_minor, real_t* out_val, unsigned long long* out_indptr,
                                size_t* nnz_out, hipStream_t stream);

namespace {

constexpr size_t LDS_RESIDENT_LIMIT = 64 * 1024;  // largest tile a single wave may claim
constexpr int TEAM_LAUNCH_MAX = 32;   // team launches per half-sweep call (one per team size: <= 22 lane-team sizes + the giant rows + the register teams' shapes)
// layout of poismf_hip_session::d_team_err, in words: [0] team launches that gave up and were re-run since the word was last read, [1] spare,
// [2 + i] set by team launch i of the current half when it gives up, [2 + TEAM_LAUNCH_MAX + i] rows launch i found unchanged (TNCG early stop:
// added to the half's counter only when the launch's results are kept)
constexpr int TEAM_ERR_WORDS = 2 + 2 * TEAM_LAUNCH_MAX;
constexpr int MAX_LAUNCHES = 256;  // launches per half-sweep call that get a row-queue head: length classes are multiples of 16 up to 256, of 64
                                   // up to 2048, then powers of two (<= 60 per segment); a call over several segments concatenates their bins

struct Bin {
    unsigned begin, count;  // range of the nnz-sorted permutation
    unsigned max_nnz;       // longest row actually in the bin (sizes tiles)
    unsigned cls;           // upper bound of the bin's length class (decides the code path: a function of the row alone)
    unsigned long long nnz; // nonzeros of the bin's rows (algorithmic bytes of a launch, bench.py's per-launch roofline)
};

struct Half {
    size_t dimM = 0, dimF = 0;
    size_t row_begin = 0, row_end = 0;  // shard
    size_t nnz = 0;
    bool x_positive = false;            // every stored value > 0 (RowParams::x_pos)
    unsigned long long* d_indptr = nullptr;
    unsigned* d_indices = nullptr;
    real_t* d_values = nullptr;
    unsigned* d_perm = nullptr;
    RowDesc* d_desc = nullptr;
    unsigned* d_eval_rows = nullptr;      // per local row: passes over its tile while profiling (allocated on demand)
    unsigned* d_dec_rows = nullptr;       // per local row: { iterations | rc << 24, evaluations } of its solver while profiling
    // The shard's rows are cut into contiguous SEGMENTS (one unless the multi-GPU driver asks for more, so that a
    // segment's rows can travel while the next one computes); within a segment rows are sorted by length and binned.
    struct Segment { unsigned row_lo, row_hi; std::vector<Bin> bins; };
    std::vector<Segment> segs;
};

struct ProfRec { hipEvent_t t0, t1; int which; };
// profiling sessions also bracket every row-bin launch (on the stream it is issued on): bench.py's per-launch roofline
struct LaunchRec { hipEvent_t t0, t1; int which; std::string name; unsigned rows; unsigned long long nnz; };

}  // namespace

struct poismf_hip_session {
    int device = 0;
    int num_cu = 256;
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
    unsigned long long* d_team = nullptr;   // team launches (plan.hpp, TEAM_*): allocated by the first one
    unsigned long long* d_gt = nullptr;     // giant-row and lane-team launches (row_eval.hpp, GT_*): one GT_BUF_BYTES area PER LAUNCH of a half, zeroed together before the first
    size_t gt_areas = 0, team_areas = 0;    // areas d_gt / d_team hold
    unsigned* d_arrive = nullptr;           // workgroups of the forked long-row launch that have started (half_sweep_impl)
    unsigned long long gate_budget = 200000;   // ticks of the wall clock the hold-back gate waits at most: 2 ms (session_alloc)
    unsigned* d_team_err = nullptr;         // TEAM_ERR_WORDS words (above): give-ups so far, and an error word and a tally per team launch of the current half
    real_t* d_team_backup = nullptr;        // the rows the team launches of a half start from, launch after launch (restored before a re-run)
    unsigned* d_team_eval_backup = nullptr; // profiling sessions: those rows' evaluation counters (a re-run must not count an abandoned launch's evaluations)
    size_t team_backup_elems = 0, team_eval_backup_rows = 0;
    bool team_launched = false;             // since the words were last read
    bool teams_off = false;                 // a team launch of this session timed out: no more multi-CU launches for the rest of it (team_check)
    int colsum_waves = 512;           // blocks (of 8 waves) in the first stage of the column sums
    bool partials_given = false;      // d_partial already holds every block's partial sum for the NEXT half-sweep (poismf_hip_session_partials_ready)
    const real_t* partials_of = nullptr;   // ... of THIS factor (the last poismf_hip_session_colsum_partial's): a half-sweep over the other factor, a factor
                                      // written since, or a half with a caller-supplied sum does not take them for its own
    bool profiling = false;
    std::vector<ProfRec> prof;
    std::vector<LaunchRec> lprof;
    std::string last_plan[2];         // the launches of the most recent half-sweep of each half, as text
};

namespace {

void free_half(Half& h, hipStream_t stream)
{
    pmf_free(h.d_indptr, stream);
    pmf_free(h.d_indices, stream);
    pmf_free(h.d_values, stream);
    pmf_free(h.d_perm, stream);
    pmf_free(h.d_desc, stream);
    pmf_free(h.d_eval_rows, stream);
    pmf_free(h.d_dec_rows, stream);
    h = Half();
}

// ---- device-side set-up of one half ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void rebase_indptr_kernel(unsigned long long* indptr, size_t n, unsigned long long base)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) indptr[i] -= base;
}
// ---- a team launch that gives up must not cost the fit (reg_eval.hpp, M_ > 1: an exchange between CUs timed out) -----------
// The rows the team launches of a half cover are saved BEFORE the half's first launch (empty chip); AFTER its last launch has ended, every team
// launch whose error word is set has its rows put back and run again on the streamed LDS kernel (both gated on that word), and one fold kernel
// settles the counters.  Rounds 2-5 bracketed every team launch with these kernels on the launch's own stream: each of them then queued for a
// wave slot behind the other stream's persistent 512-register workgroups (round 5's profile: a restore kernel whose body is one compare, 13 ms
// on average, 8 per C5 sweep) and held the next team launch back.  Now a team launch is ONE dispatch and the healthy case pays its
// bookkeeping where nothing else is resident.
__global__ __launch_bounds__(256) void team_save_rows_kernel(const real_t* M, const RowDesc* desc, unsigned nrows, unsigned row_offset, int k, real_t* backup,
                                                             const unsigned* eval_rows, unsigned* eval_backup)
{
    const size_t n = (size_t)nrows * (size_t)k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (size_t)k, c = i % (size_t)k;
        backup[i] = M[(size_t)(row_offset + desc[r].lrow) * (size_t)k + c];
        if (eval_rows != nullptr && c == 0) eval_backup[r] = eval_rows[desc[r].lrow];
    }
}
__global__ __launch_bounds__(256) void team_restore_rows_kernel(real_t* M, real_t* Mp, int ldM, const RowDesc* desc, unsigned nrows, unsigned row_offset,
                                                                int k, const real_t* backup, const unsigned* err, unsigned* eval_rows, const unsigned* eval_backup)
{
    if (*err == 0) return;
    const size_t n = (size_t)nrows * (size_t)k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (size_t)k, c = i % (size_t)k;
        const size_t row = (size_t)(row_offset + desc[r].lrow);
        M[row * (size_t)k + c] = backup[i];
        if (Mp != nullptr) Mp[row * (size_t)ldM + c] = backup[i];
        if (eval_rows != nullptr && c == 0) eval_rows[desc[r].lrow] = eval_backup[r];
    }
}
// After the re-runs: a launch that gave up is counted ([0]) and its own tally of unchanged rows dropped (its re-run counted them again, straight
// into the half's counter, ref: src/poismf.c:393-403); a launch that kept its results adds its tally.  The per-launch words are left zeroed.
__global__ void team_fold_kernel(unsigned* err, int n, unsigned* n_unchanged)
{
    for (int i = 0; i < n; i++) {
        if (err[2 + i] != 0) err[0] += 1;
        else if (n_unchanged != nullptr) *n_unchanged += err[2 + TEAM_LAUNCH_MAX + i];
        err[2 + i] = 0;
        err[2 + TEAM_LAUNCH_MAX + i] = 0;
    }
}

// The hold-back of a half-sweep's other bins behind its forked long-row launch (poismf_hip_half_sweep): one wave that returns when `goal`
// workgroups of that launch have counted themselves in -- or when `budget` ticks of the constant-rate wall clock have passed, whichever is
// first.  A BOUNDED wait on purpose: rounds 3-4a held the stream with hipStreamWaitValue32, and under `rocprofv3 --pmc` (dispatches serialised
// by the tool) that wait kept the long-row launch from ever starting -- a counter pass over config C5 sat there until gpurun's limit
// (one hour of round 4's GPU time).  This kernel gives up after 2 ms and the bins then run one after the other.
__global__ void hold_back_gate_kernel(const unsigned* word, unsigned goal, unsigned long long budget)
{
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < goal) {
        if (wall_clock64() - t0 > budget) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

// flag |= 1 when some stored value is not > 0 (zero, negative, NaN)
__global__ __launch_bounds__(256) void values_positive_kernel(const real_t* v, size_t n, unsigned* flag)
{
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad = bad || !(v[i] > (real_t)0);
    if (bad) atomicOr(flag, 1u);
}
__global__ __launch_bounds__(256) void row_desc_kernel(const unsigned long long* indptr, const unsigned* perm, size_t n, RowDesc* desc)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned r = perm[i];
        const unsigned long long p0 = indptr[r];
        desc[i] = { (unsigned)p0, (unsigned)(p0 >> 32), (unsigned)(indptr[r + 1] - p0), r };
    }
}

// bin classes: multiples of 16 up to 256 nonzeros, multiples of 64 up to 2048 (the hand-overs between 1, 2, 4
// and 8 waves per row and between the team shapes fall on those), then powers of two.  The LDS tile of a launch is sized by the longest
// row of its bin, and LDS is what limits the waves per CU, so fine classes where most rows live buy
// occupancy (C2: 100 +- 10 nnz per row -> 7 waves per CU instead of 5).  Which ENGINE a row takes, and how
// many waves share it, is decided by the class bound alone -- never by which other rows happen to be in the
// shard -- so a row's arithmetic does not depend on how the matrix is cut into shards.
unsigned length_class(unsigned n)
{
    if (n <= 256) return std::max(16u, (n + 15u) / 16u * 16u);
    if (n <= 2048) return (n + 63u) / 64u * 64u;   // (up to the longest team row: every class maps to ONE team shape, plan.hpp team_shape_for)
    unsigned cls = 4096;
    while (cls < n) cls <<= 1;
    return cls;
}

// h.d_indptr (shard-local, nloc + 1), h.d_indices, h.d_values are on the device: per segment, sort the rows by length
// (longest first, equal lengths in row order), build the row descriptors on the device and the length bins on the host
// (from the sorted lengths: 4 bytes per row come back over PCIe, the only host work left is one linear scan).
// In two steps, so that a caller can put other work (run_poismf: the factors' upload) between the launches and the one place that
// waits for them.
struct HalfPending { unsigned* d_len = nullptr; unsigned* d_flag = nullptr; };
int finish_half_launch(Half& h, hipStream_t stream, HalfPending& pend, int nseg = 1)
{
    const size_t nloc = h.row_end - h.row_begin;
    if (h.d_perm == nullptr) HIP_TRY(pmf_alloc(&h.d_perm, sizeof(unsigned) * (nloc ? nloc : 1), stream));
    if (h.d_desc == nullptr) HIP_TRY(pmf_alloc(&h.d_desc, sizeof(RowDesc) * (nloc ? nloc : 1), stream));
    h.segs.clear();
    nseg = std::max(nseg, 1);   // as asked (segments of a short shard may be empty): every rank cuts its shard the same way
    if (nloc == 0) { for (int j = 0; j < nseg; j++) h.segs.push_back({ 0u, 0u, {} }); return 0; }
    HIP_TRY(pmf_alloc(&pend.d_len, sizeof(unsigned) * nloc, stream));
    for (int j = 0; j < nseg; j++) {
        const size_t lo = nloc * (size_t)j / (size_t)nseg, hi = nloc * (size_t)(j + 1) / (size_t)nseg;   // == dist.segment_of
        h.segs.push_back({ (unsigned)lo, (unsigned)hi, {} });
        if (hi > lo && poismf_hip_device_sort_rows(h.d_indptr + lo, hi - lo, (unsigned)lo, h.d_perm + lo, pend.d_len + lo, stream)) {
            pmf_free(pend.d_len, stream);
            pend.d_len = nullptr;
            return 1;
        }
    }
    const unsigned grid = (unsigned)std::min<size_t>((nloc + 255) / 256, 2048);
    hipLaunchKernelGGL(row_desc_kernel, dim3(grid), dim3(256), 0, stream, h.d_indptr, h.d_perm, nloc, h.d_desc);
    // are all stored values positive (Poisson counts)?  One pass over the values, the flag rides in front of the lengths
    HIP_TRY(pmf_alloc(&pend.d_flag, sizeof(unsigned), stream));
    HIP_TRY(hipMemsetAsync(pend.d_flag, 0, sizeof(unsigned), stream));
    if (h.nnz > 0)
        hipLaunchKernelGGL(values_positive_kernel, dim3((unsigned)std::min<size_t>((h.nnz + 255) / 256, 2048)), dim3(256), 0, stream, h.d_values, h.nnz, pend.d_flag);
    return 0;
}
int finish_half_collect(Half& h, hipStream_t stream, HalfPending& pend)
{
    const size_t nloc = h.row_end - h.row_begin;
    if (nloc == 0) return 0;
    std::vector<unsigned> len(nloc);
    hipError_t e = pmf_download(len.data(), pend.d_len, sizeof(unsigned) * nloc, stream);
    unsigned not_positive = 1;
    if (e == hipSuccess) e = pmf_download(&not_positive, pend.d_flag, sizeof(unsigned), stream);
    h.x_positive = not_positive == 0;
    pmf_free(pend.d_flag, stream);
    pmf_free(pend.d_len, stream);
    pend = HalfPending();
    HIP_TRY(e);
    for (auto& sg : h.segs) {
        for (size_t i = sg.row_lo; i < sg.row_hi; i++) {
            const unsigned cls = length_class(len[i]);
            if (sg.bins.empty() || cls != sg.bins.back().cls) sg.bins.push_back({ (unsigned)i, 0u, len[i], cls, 0ull });   // sorted: the first row of a bin is its longest
            sg.bins.back().count++;
            sg.bins.back().nnz += len[i];
        }
    }
    return 0;
}
int finish_half(Half& h, hipStream_t stream, int nseg = 1)
{
    HalfPending pend;
    if (finish_half_launch(h, stream, pend, nseg)) { pmf_free(pend.d_flag, stream); pmf_free(pend.d_len, stream); return 1; }
    return finish_half_collect(h, stream, pend);
}

// Upload rows [r0, r1) of a host CSR (size_t indices) with shard-local pointers; the indices are narrowed to u32 on
// the device (the binding rejects dimensions above INT_MAX, ref: poismf_c_wrapper.pxi:78-80).
// (pend != nullptr: the row sort is launched but not waited for -- the caller owes a finish_half_collect)
int build_half(Half& h, hipStream_t stream, const real_t* val, const sparse_ix* indptr, const sparse_ix* indices,
               size_t dimM, size_t dimF, size_t r0, size_t r1, int device = 0, HalfPending* pend = nullptr)
{
    h.dimM = dimM; h.dimF = dimF; h.row_begin = r0; h.row_end = r1;
    const size_t nloc = r1 - r0;
    const size_t base = (size_t)indptr[r0];
    h.nnz = (size_t)indptr[r1] - base;
    HIP_TRY(pmf_alloc(&h.d_indptr, sizeof(unsigned long long) * (nloc + 1), stream));
    HIP_TRY(pmf_alloc(&h.d_indices, sizeof(unsigned) * (h.nnz ? h.nnz : 1), stream));
    HIP_TRY(pmf_alloc(&h.d_values, sizeof(real_t) * (h.nnz ? h.nnz : 1), stream));
    pmf_tl("half: device arrays allocated");
    if constexpr (sizeof(sparse_ix) == sizeof(unsigned long long)) {
        // C / Python ABI: size_t indices go up as they are and are narrowed to u32 by a kernel
        HIP_TRY(pmf_upload(h.d_indptr, indptr + r0, sizeof(unsigned long long) * (nloc + 1), stream));
        // indices: narrowed to u32 by host threads on their way into pinned chunks (half the bytes over PCIe, devmem.hpp); the
        // plain path -- whole size_t array up, narrowed by a kernel -- when the staged one is not available
        hipError_t se = hipErrorNotReady;
        if (h.nnz) {
            const sparse_ix* src = indices + base;
            se = pmf_upload_staged(h.d_indices, h.nnz, sizeof(unsigned), device, stream, [src](void* pin, size_t i0, size_t cnt) {
                unsigned* o = (unsigned*)pin;
                const sparse_ix* q = src + i0;
                for (size_t i = 0; i < cnt; i++) o[i] = (unsigned)q[i];
            });
            if (se != hipSuccess && se != hipErrorNotReady) HIP_TRY(se);
            pmf_tl("half: row pointers up, indices narrowed and handed to the DMA queue");
        }
        if (h.nnz && se != hipSuccess) {
            unsigned long long* d_wide = nullptr;
            HIP_TRY(pmf_alloc(&d_wide, sizeof(unsigned long long) * h.nnz, stream));
            hipError_t e = pmf_upload(d_wide, indices + base, sizeof(unsigned long long) * h.nnz, stream);
            if (e == hipSuccess && poismf_hip_device_narrow(d_wide, h.nnz, h.d_indices, stream)) e = hipErrorUnknown;
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            pmf_free(d_wide, stream);
            HIP_TRY(e);
        }
    } else {
        // R ABI: int indices are the device's width already; the row pointers are widened on the host (dim + 1 values)
        std::vector<unsigned long long> wide(nloc + 1);
        for (size_t i = 0; i <= nloc; i++) wide[i] = (unsigned long long)indptr[r0 + i];
        HIP_TRY(pmf_upload(h.d_indptr, wide.data(), sizeof(unsigned long long) * (nloc + 1), stream));
        HIP_TRY(pmf_upload(h.d_indices, indices + base, sizeof(unsigned) * h.nnz, stream));
    }
    if (base != 0) {
        hipLaunchKernelGGL(rebase_indptr_kernel, dim3((unsigned)std::min<size_t>((nloc + 256) / 256, 2048)), dim3(256), 0, stream, h.d_indptr, nloc + 1,
                           (unsigned long long)base);
    }
    {
        const real_t* src = val + base;
        const hipError_t se = pmf_upload_staged(h.d_values, h.nnz, sizeof(real_t), device, stream, [src](void* pin, size_t i0, size_t cnt) {
            memcpy(pin, src + i0, cnt * sizeof(real_t));
        });
        if (se == hipErrorNotReady) HIP_TRY(pmf_upload(h.d_values, val + base, sizeof(real_t) * h.nnz, stream));
        else HIP_TRY(se);
    }
    pmf_tl("half: values handed to the DMA queue");
    if (pend != nullptr) {
        if (finish_half_launch(h, stream, *pend)) { pmf_free(pend->d_flag, stream); pmf_free(pend->d_len, stream); *pend = HalfPending(); return 1; }
        return 0;
    }
    const int bad = finish_half(h, stream);
    pmf_tl("half: rows sorted, lengths back, bins cut");
    return bad;
}

bool prefetch_enabled() { return true; }   // (streamed rows request the next chunk's tile a chunk ahead: row_eval.hpp, PF)

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
    // Only for streamed rows: there every line-search trial would otherwise be a fresh gather from L2/HBM
    // (C3 B half: 247 -> 146 ms).  For LDS-resident rows a trial is a cheap pass over the tile already and the
    // cached variant buys nothing: it halves the passes over the tile (C2 CG fp64: 20 -> 10 per row) and the sweep takes
    // the same 10.3 ms -- those rows are bound by the solver's chain of k-vector reductions and scalar decisions at one
    // wave per SIMD, not by the tile passes.
    if (want_pq && !g.resident && (size_t)2 * bin_max_nnz * sizeof(real_t) <= 48 * 1024)
        g.pq_cap = (int)((bin_max_nnz + 15u) / 16u * 16u);
    g.prefetch = (!g.resident && prefetch_enabled()) ? 1 : 0;
    return g;
}

}  // namespace

namespace {

// column-sum kernels: elements per lane in the plain lane <-> element layout
int nc_for_k(size_t k) { return k <= 64 ? 1 : (k <= 128 ? 2 : (k <= 256 ? 4 : (k <= 512 ? 8 : 0))); }
inline int colsum_blocks_for(const poismf_hip_session* s, size_t n)
{
    return (int)std::min<size_t>((size_t)s->colsum_waves, std::max<size_t>((n + COLSUM_BLOCK_WAVES - 1) / COLSUM_BLOCK_WAVES, 1));
}
template <int NC> int launch_colsum_partial(poismf_hip_session* s, const real_t* M, size_t n, int b_lo, int b_hi)
{
    const int nw = colsum_blocks_for(s, n);
    if (b_lo < 0 || b_hi > nw || b_lo > b_hi) return 1;
    s->partials_of = M;
    s->partials_given = false;
    if (b_hi > b_lo)
        hipLaunchKernelGGL((colsum_partial_kernel<real_t, NC>), dim3(b_hi - b_lo), dim3(WAVE * COLSUM_BLOCK_WAVES), 0, s->stream, M, n, (int)s->k,
                           s->d_partial, (unsigned)b_lo, (unsigned)nw);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <int NC> int launch_colsum(poismf_hip_session* s, const real_t* M, size_t n, real_t l1, real_t scale, int nscale)
{
    const int nw = colsum_blocks_for(s, n);
    // first stage: all blocks here, or only [b_lo, b_hi) (the others are the peers' and have been put into d_partial by the caller), or none
    int b_lo = 0, b_hi = nw;
    if (s->partials_given && s->partials_of == M) { b_lo = b_hi = 0; }
    s->partials_given = false;        // (one shot, whoever consumes or declines it)
    if (b_hi > b_lo)
        hipLaunchKernelGGL((colsum_partial_kernel<real_t, NC>), dim3(b_hi - b_lo), dim3(WAVE * COLSUM_BLOCK_WAVES), 0, s->stream, M, n, (int)s->k,
                           s->d_partial, (unsigned)b_lo, (unsigned)nw);
    hipLaunchKernelGGL((colsum_final_kernel<real_t, NC>), dim3(1), dim3(WAVE * COLSUM_FINAL_WAVES), 0, s->stream, s->d_partial, nw, (int)s->k, l1,
                       scale, nscale, s->d_bsum);
    HIP_TRY(hipGetLastError());
    return 0;
}

int colsum_partial(poismf_hip_session* s, const real_t* M, size_t n, int b_lo, int b_hi)
{
    switch (nc_for_k(s->k)) {
        case 1: return launch_colsum_partial<1>(s, M, n, b_lo, b_hi);
        case 2: return launch_colsum_partial<2>(s, M, n, b_lo, b_hi);
        case 4: return launch_colsum_partial<4>(s, M, n, b_lo, b_hi);
        case 8: return launch_colsum_partial<8>(s, M, n, b_lo, b_hi);
    }
    return 1;
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

extern "C" {

// Streams are recycled across sessions: creating one costs 7.5 ms on this stack (scripts/probes/h2d_probe.hip) -- with two
// per run_poismf call that was most of the call's set-up time on config C2.  Idle streams wait here, per device.
static std::mutex g_stream_mutex;
static std::vector<hipStream_t> g_idle_streams[64];
static int cached_stream(int device, hipStream_t* out)
{
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lk(g_stream_mutex);
        auto& v = g_idle_streams[device];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return 0; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking) != hipSuccess;
}
static void release_stream(int device, hipStream_t st)
{
    (void)hipStreamSynchronize(st);
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lk(g_stream_mutex);
        if (g_idle_streams[device].size() < 8) { g_idle_streams[device].push_back(st); return; }
    }
    (void)hipStreamDestroy(st);
}

// Everything of a session except the two halves of X: streams, the replicated factors (+ their line-padded gather
// copies), column-sum scratch.  On failure the partly built session is destroyed and nullptr returned.
static poismf_hip_session* session_alloc(int device, void* stream, size_t dimA, size_t dimB, size_t k)
{
    if (k == 0 || slots_per_lane(k) == 0 || nc_for_k(k) == 0) {
        fprintf(stderr, "poismf_hip: k = %zu is outside the supported range (1..%d)\n", k, 128 * SLOT_ELEMS);
        return nullptr;
    }
    if (const hipError_t e = hipSetDevice(device); e != hipSuccess) {   // no device, wrong index: rc 1, and stderr says it was not memory
        pmf_last_hip_error() = e;
        return nullptr;
    }
    poismf_hip_session* s = new (std::nothrow) poismf_hip_session();
    if (!s) return nullptr;
    s->device = device;
    {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) s->num_cu = n;
    }
    {
        int khz = 0;   // constant-rate counter behind wall_clock64(): 100 MHz on this part
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) s->gate_budget = 2ull * (unsigned long long)khz;
    }
    s->stream = (hipStream_t)stream;
    auto fail = [&]() -> poismf_hip_session* { poismf_hip_session_destroy(s); return nullptr; };
    if (s->stream == nullptr) {  // no stream given: the session owns a non-blocking stream (NOT the legacy default stream)
        if (cached_stream(device, &s->stream)) { delete s; return nullptr; }
        s->owns_stream = true;
    }
    s->dimA = dimA; s->dimB = dimB; s->k = k;
    // behind each factor: one all-zero row (the register engine points the unused steps of a row at it) + 16 B so that
    // the last 16-byte slot of the last row stays in bounds
    const size_t slack = k * sizeof(real_t) + 16;
    if (pmf_alloc(&s->dA, dimA * k * sizeof(real_t) + slack, s->stream) != hipSuccess) return fail();
    if (pmf_alloc(&s->dB, dimB * k * sizeof(real_t) + slack, s->stream) != hipSuccess) return fail();
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
            if (pmf_alloc(&s->dAp, dimA * padb + pslack, s->stream) != hipSuccess) return fail();
            if (pmf_alloc(&s->dBp, dimB * padb + pslack, s->stream) != hipSuccess) return fail();
            if (hipMemsetAsync(s->dAp, 0, dimA * padb + pslack, s->stream) != hipSuccess) return fail();
            if (hipMemsetAsync(s->dBp, 0, dimB * padb + pslack, s->stream) != hipSuccess) return fail();
        }
    }
    if (pmf_alloc(&s->d_bsum, k * sizeof(real_t) + slack, s->stream) != hipSuccess) return fail();
    if (pmf_alloc(&s->d_partial, (size_t)s->colsum_waves * k * sizeof(real_t), s->stream) != hipSuccess) return fail();
    if (pmf_alloc(&s->d_counter, sizeof(unsigned), s->stream) != hipSuccess) return fail();
    if (pmf_alloc(&s->d_queue, sizeof(unsigned) * (MAX_LAUNCHES + 2 * TEAM_LAUNCH_MAX), s->stream) != hipSuccess) return fail();   // (+: a head of its own for every team launch of a half, and one for its streamed re-run)
    if (pmf_alloc(&s->d_arrive, sizeof(unsigned), s->stream) != hipSuccess) return fail();
    if (pmf_alloc(&s->d_team_err, TEAM_ERR_WORDS * sizeof(unsigned), s->stream) != hipSuccess) return fail();
    if (hipMemsetAsync(s->d_team_err, 0, TEAM_ERR_WORDS * sizeof(unsigned), s->stream) != hipSuccess) return fail();
    if (cached_stream(device, &s->aux_stream)) return fail();
    if (hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess) return fail();
    if (hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess) return fail();
    return s;
}

int poismf_hip_session_create(poismf_hip_session** out, int device, void* stream, const real_t* Xr,
                              const sparse_ix* Xr_indptr, const sparse_ix* Xr_indices, const real_t* Xc,
                              const sparse_ix* Xc_indptr, const sparse_ix* Xc_indices, size_t dimA, size_t dimB, size_t k,
                              size_t rowA_begin, size_t rowA_end, size_t rowB_begin, size_t rowB_end)
{
    *out = nullptr;
    if (rowA_end > dimA || rowB_end > dimB || rowA_begin > rowA_end || rowB_begin > rowB_end) return 1;
    poismf_hip_session* s = session_alloc(device, stream, dimA, dimB, k);
    if (!s) return 1;
    auto fail = [&]() { poismf_hip_session_destroy(s); return 1; };
    // half 0 updates B: rows of the CSC; half 1 updates A: rows of the CSR
    if (Xc_indptr != nullptr &&
        build_half(s->half[0], s->stream, Xc, Xc_indptr, Xc_indices, dimB, dimA, rowB_begin, rowB_end, device)) return fail();
    if (build_half(s->half[1], s->stream, Xr, Xr_indptr, Xr_indices, dimA, dimB, rowA_begin, rowA_end, device)) return fail();
    *out = s;
    return 0;
}

// One orientation of a device COO -> the half's own (exactly sized) CSR arrays, rows [m0, m1) only.
static int half_from_device_coo(Half& h, hipStream_t stream, const unsigned* d_major, const unsigned* d_minor, const real_t* d_val, size_t n,
                                size_t dimM, size_t dimF, size_t m0, size_t m1)
{
    h.dimM = dimM; h.dimF = dimF; h.row_begin = m0; h.row_end = m1;
    const size_t nloc = m1 - m0;
    unsigned* t_idx = nullptr;
    real_t* t_val = nullptr;
    HIP_TRY(pmf_alloc(&h.d_indptr, sizeof(unsigned long long) * (nloc + 1), stream));
    hipError_t e = pmf_alloc(&t_idx, sizeof(unsigned) * n, stream);
    if (e == hipSuccess) e = pmf_alloc(&t_val, sizeof(real_t) * n, stream);
    size_t uniq = 0;
    int rc = e != hipSuccess;
    if (!rc) rc = poismf_hip_device_coo_to_cs(d_major, d_minor, d_val, n, m0, m1, t_idx, t_val, h.d_indptr, &uniq, stream);
    if (!rc) {
        h.nnz = uniq;
        // the conversion's outputs have room for all n triplets; the session keeps exactly sized copies
        if (pmf_alloc(&h.d_indices, sizeof(unsigned) * (uniq ? uniq : 1), stream) != hipSuccess ||
            pmf_alloc(&h.d_values, sizeof(real_t) * (uniq ? uniq : 1), stream) != hipSuccess ||
            hipMemcpyAsync(h.d_indices, t_idx, sizeof(unsigned) * uniq, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
            hipMemcpyAsync(h.d_values, t_val, sizeof(real_t) * uniq, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess)
            rc = 1;
    }
    pmf_free(t_idx, stream);
    pmf_free(t_val, stream);
    return rc ? 1 : finish_half(h, stream);
}

int poismf_hip_session_create_coo(poismf_hip_session** out, int device, void* stream, const sparse_ix* row, const sparse_ix* col,
                                  const real_t* val, size_t n, size_t dimA, size_t dimB, size_t k, size_t rowA_begin,
                                  size_t rowA_end, size_t rowB_begin, size_t rowB_end)
{
    *out = nullptr;
    if (n == 0 || n > 0xffffffffull || dimA > 0x7fffffffull || dimB > 0x7fffffffull) return 1;
    if (rowA_end > dimA || rowB_end > dimB || rowA_begin > rowA_end || rowB_begin > rowB_end) return 1;
    poismf_hip_session* s = session_alloc(device, stream, dimA, dimB, k);
    if (!s) return 1;
    unsigned *d_row = nullptr, *d_col = nullptr;
    real_t* d_val = nullptr;
    int rc = 1;
    do {
        if (pmf_alloc(&d_row, sizeof(unsigned) * n, s->stream) != hipSuccess || pmf_alloc(&d_col, sizeof(unsigned) * n, s->stream) != hipSuccess ||
            pmf_alloc(&d_val, sizeof(real_t) * n, s->stream) != hipSuccess)
            break;
        {
            std::vector<unsigned> h32;
            try { h32.resize(n); } catch (const std::bad_alloc&) { break; }
            // (an index outside the matrix -- or a negative one reinterpreted as size_t -- would become a gather offset into the
            // factors: rc 3, which the binding turns into ValueError)
            bool bad_index = false;
            for (size_t i = 0; i < n; i++) { bad_index |= (size_t)row[i] >= dimA; h32[i] = (unsigned)row[i]; }
            if (bad_index) { rc = 3; break; }
            if (hipMemcpy(d_row, h32.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice) != hipSuccess) break;
            for (size_t i = 0; i < n; i++) { bad_index |= (size_t)col[i] >= dimB; h32[i] = (unsigned)col[i]; }
            if (bad_index) { rc = 3; break; }
            if (hipMemcpy(d_col, h32.data(), sizeof(unsigned) * n, hipMemcpyHostToDevice) != hipSuccess) break;
        }
        if (hipMemcpy(d_val, val, sizeof(real_t) * n, hipMemcpyHostToDevice) != hipSuccess) break;
        // half 0 updates B: the CSC (major = column); half 1 updates A: the CSR (major = row)
        if (half_from_device_coo(s->half[0], s->stream, d_col, d_row, d_val, n, dimB, dimA, rowB_begin, rowB_end)) break;
        if (half_from_device_coo(s->half[1], s->stream, d_row, d_col, d_val, n, dimA, dimB, rowA_begin, rowA_end)) break;
        rc = 0;
    } while (0);
    pmf_free(d_row, s->stream);
    pmf_free(d_col, s->stream);
    pmf_free(d_val, s->stream);
    if (rc) { poismf_hip_session_destroy(s); return rc; }
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
    for (auto& p : s->lprof) { (void)hipEventDestroy(p.t0); (void)hipEventDestroy(p.t1); }
    s->lprof.clear();
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    const hipStream_t own = s->owns_stream ? s->stream : nullptr, aux = s->aux_stream;
    free_half(s->half[0], s->stream);
    free_half(s->half[1], s->stream);
    pmf_free(s->dA, s->stream);
    pmf_free(s->dB, s->stream);
    pmf_free(s->dAp, s->stream);
    pmf_free(s->dBp, s->stream);
    pmf_free(s->d_bsum, s->stream);
    pmf_free(s->d_partial, s->stream);
    pmf_free(s->d_counter, s->stream);
    pmf_free(s->d_queue, s->stream);
    pmf_free(s->d_gt, s->stream);
    pmf_free(s->d_team, s->stream);
    pmf_free(s->d_team_err, s->stream);
    pmf_free(s->d_arrive, s->stream);
    pmf_free(s->d_team_backup, s->stream);
    pmf_free(s->d_team_eval_backup, s->stream);
    (void)hipStreamSynchronize(s->stream);   // the stream-ordered frees have run
    if (aux) release_stream(s->device, aux);
    if (own) release_stream(s->device, own);
    delete s;
}

// the device arrays kept from finished sessions (devmem.hpp) go back to the driver
void poismf_hip_release_cache(void) { pmf_release_cache(); }
// how much released device memory may be kept for the next call (MB; 0 = nothing, the default); returns the previous limit
size_t poismf_hip_set_device_cache_mb(size_t mb) { return pmf_set_cache_limit_mb(mb); }

// Whoever asks for the device pointers may write through them: the padded gather copies are re-derived afterwards.
real_t* poismf_hip_session_A(poismf_hip_session* s) { s->padded_fresh[1] = false; return s->dA; }
real_t* poismf_hip_session_B(poismf_hip_session* s) { s->padded_fresh[0] = false; return s->dB; }
// ... and whoever keeps such a pointer says so after every later write (which = 0: B was written, 1: A)
void poismf_hip_session_factors_dirty(poismf_hip_session* s, int which)
{
    s->padded_fresh[which ? 1 : 0] = false;
    if (s->partials_of == (which ? s->dA : s->dB)) { s->partials_given = false; s->partials_of = nullptr; }   // partial sums of a factor written since
}

// ---- the first stage of the column sums, shared between the ranks of a multi-GPU run (ref: src/poismf.c:77-83) -----------------
// The sum over the fixed factor of half `which` (A for the B half, B for the A half) is cut into poismf_hip_session_colsum_blocks() blocks
// whose partial sums do not depend on who computes them.  A rank computes blocks [b_lo, b_hi) into the session's partial array
// (poismf_hip_session_partials: [blocks x k] real_t, device memory), receives the other blocks from its peers into the same array, and says so
// (poismf_hip_session_partials_ready): the next half-sweep then runs the fixed-order second stage only.  Same bits as the unsharded sum.
int poismf_hip_session_colsum_blocks(poismf_hip_session* s, int which) { return colsum_blocks_for(s, which ? s->dimB : s->dimA); }
int poismf_hip_session_colsum_partial(poismf_hip_session* s, int which, int b_lo, int b_hi)
{
    HIP_TRY(hipSetDevice(s->device));
    return colsum_partial(s, which ? s->dB : s->dA, which ? s->dimB : s->dimA, b_lo, b_hi);
}
real_t* poismf_hip_session_partials(poismf_hip_session* s) { return s->d_partial; }
void poismf_hip_session_partials_ready(poismf_hip_session* s) { s->partials_given = s->partials_of != nullptr; }
void* poismf_hip_session_stream(poismf_hip_session* s) { return (void*)s->stream; }
size_t poismf_hip_session_nnz(poismf_hip_session* s, int which) { return s->half[which ? 1 : 0].nnz; }

#ifdef PMF_TIMING
// development-only export (not in the header): read and reset the phase timers (they live in the row-kernel translation unit)
}
int pmf_read_timing(unsigned long long* out);
extern "C" {
__attribute__((visibility("default"))) void poismf_hip_debug_timing(unsigned long long* out) { (void)pmf_read_timing(out); }
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
    hipLaunchKernelGGL(selftest_log_kernel, dim3(256 * 8), dim3(256), 0, 0, (unsigned long long)n, d_w, d_m);
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
    HIP_TRY(pmf_upload_big(s->dA, A_host, s->dimA * s->k * sizeof(real_t), s->device, s->stream));
    HIP_TRY(pmf_upload_big(s->dB, B_host, s->dimB * s->k * sizeof(real_t), s->device, s->stream));
    s->padded_fresh[0] = s->padded_fresh[1] = false;
    s->partials_given = false; s->partials_of = nullptr;
    return 0;
}

// Team launches (reg_eval.hpp, M_ > 1) give up instead of hanging when an exchange between CUs times out; the word they set
// is read where the session synchronises anyway.  Nonzero: the factors are not to be trusted.
static int team_check(poismf_hip_session* s)
{
    if (!s->team_launched) return 0;
    unsigned w[2] = { 0, 0 };
    HIP_TRY(pmf_download(w, s->d_team_err, 2 * sizeof(unsigned), s->stream));
    s->team_launched = false;
    if (w[0] != 0) {
        // (somebody else holds CUs: every further team launch would sit out its time-out as well -- 300 ms each, up to eight per half on config C5.
        // The rest of this session plans without teams: the rows take the streamed kernels directly, which is what a re-run gives them anyway.)
        s->teams_off = true;
        fprintf(stderr, "poismf_hip: %u multi-CU row launch(es) timed out waiting for a partner CU and were re-run on the streamed path "
                        "(results are valid; another process or kernel is holding CUs; no further multi-CU launches in this session)\n", w[0]);
        HIP_TRY(hipMemsetAsync(s->d_team_err, 0, 2 * sizeof(unsigned), s->stream));
    }
    return 0;
}

int poismf_hip_session_get_factors(poismf_hip_session* s, real_t* A_host, real_t* B_host)
{
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(pmf_download_big(A_host, s->dA, s->dimA * s->k * sizeof(real_t), s->device, s->stream));
    HIP_TRY(pmf_download_big(B_host, s->dB, s->dimB * s->k * sizeof(real_t), s->device, s->stream));
    return team_check(s);
}

void poismf_hip_session_profile(poismf_hip_session* s, int enable)
{
    (void)hipStreamSynchronize(s->stream);
    for (auto& p : s->prof) { (void)hipEventDestroy(p.t0); (void)hipEventDestroy(p.t1); }
    s->prof.clear();
    for (auto& p : s->lprof) { (void)hipEventDestroy(p.t0); (void)hipEventDestroy(p.t1); }
    s->lprof.clear();
    s->profiling = enable != 0;
    for (Half& h : s->half) {
        const size_t n = h.row_end - h.row_begin;
        if (s->profiling && h.d_eval_rows == nullptr && n > 0 && pmf_alloc(&h.d_eval_rows, sizeof(unsigned) * n, s->stream) != hipSuccess) h.d_eval_rows = nullptr;
        if (h.d_eval_rows != nullptr) (void)hipMemsetAsync(h.d_eval_rows, 0, sizeof(unsigned) * n, s->stream);
        if (s->profiling && h.d_dec_rows == nullptr && n > 0 && pmf_alloc(&h.d_dec_rows, 2 * sizeof(unsigned) * n, s->stream) != hipSuccess) h.d_dec_rows = nullptr;
        if (h.d_dec_rows != nullptr) (void)hipMemsetAsync(h.d_dec_rows, 0, 2 * sizeof(unsigned) * n, s->stream);
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
    std::vector<unsigned long long> ptr(n + 1);
    HIP_TRY(hipMemcpy(ev.data(), h.d_eval_rows, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ptr.data(), h.d_indptr, sizeof(unsigned long long) * (n + 1), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        *tile_passes += ev[i];
        *nnz_passes += (unsigned long long)ev[i] * (ptr[i + 1] - ptr[i]);
    }
    return 0;
}

// The decisions of the solver on every row of half `which` in the most recent half-sweep of a profiling session: out[2 r] =
// iterations | rc << 24, out[2 r + 1] = evaluations, counted as the reference's minimize_nonneg_cg / tnc count them (local row r).
int poismf_hip_session_decisions(poismf_hip_session* s, int which, unsigned* out, size_t nrows)
{
    Half& h = s->half[which ? 1 : 0];
    if (h.d_dec_rows == nullptr) return 1;
    HIP_TRY(pmf_download(out, h.d_dec_rows, 2 * sizeof(unsigned) * std::min(nrows, h.row_end - h.row_begin), s->stream));
    return 0;
}

// Sums over the rows of half `which` of what the solvers decided in the most recent half-sweep of a profiling session, for the
// flop count the reference's arithmetic would need for the same decisions ("Flops"): out[0] = sum of iterations,
// out[1] = sum of evaluations, out[2] = sum of nnz x iterations, out[3] = sum of nnz x evaluations.
int poismf_hip_session_decision_stats(poismf_hip_session* s, int which, unsigned long long* out)
{
    HIP_TRY(hipStreamSynchronize(s->stream));
    Half& h = s->half[which ? 1 : 0];
    const size_t n = h.row_end - h.row_begin;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (h.d_dec_rows == nullptr) return 1;
    if (n == 0) return 0;
    std::vector<unsigned> dec(2 * n);
    std::vector<unsigned long long> ptr(n + 1);
    HIP_TRY(hipMemcpy(dec.data(), h.d_dec_rows, 2 * sizeof(unsigned) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ptr.data(), h.d_indptr, sizeof(unsigned long long) * (n + 1), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        const unsigned long long it = dec[2 * i] & 0xffffffu, ev = dec[2 * i + 1], nz = ptr[i + 1] - ptr[i];
        out[0] += it; out[1] += ev; out[2] += nz * it; out[3] += nz * ev;
    }
    return 0;
}

// bsum_override != nullptr: use this HOST k-vector (already carrying l1 and any PG scaling) instead of the column
// sums of the fixed factor; neg_step_override then replaces -step_size as the PG scale of the per-row Bsum_w.
// seg < 0: every segment of the shard; seg >= 0: that segment only -- segment 0 then also runs the prologue (column sums,
// refresh of the padded gather copy, reset of the early-stop counter), and the counter is read by whichever call passes
// n_unchanged (the last segment).
static int half_sweep_impl(poismf_hip_session* s, int which, const poismf_hip_params* p, real_t step_size, real_t cnst_div,
                           size_t* n_unchanged, const real_t* bsum_override, real_t neg_step_override, real_t neg_step2 = (real_t)1,
                           int seg = -1)
{
    HIP_TRY(hipSetDevice(s->device));
    which = which ? 1 : 0;
    Half& h = s->half[which];
    real_t* M = which ? s->dA : s->dB;
    const real_t* F = which ? s->dB : s->dA;
    const size_t dimF = which ? s->dimB : s->dimA;
    const bool is_pg = p->method == POISMF_PG;
    const int pm = p->method == POISMF_EVAL ? POISMF_CG : p->method;   // the evaluation-only kernels (plan.hpp, K_EVAL) are planned like CG
    const bool weighted = p->w_mult != (real_t)1.;
    if (seg >= (int)h.segs.size()) return 1;
    const bool prologue = seg <= 0;

    // column sums of the fixed factor (+ l1), with the PG pre-scaling when w == 1:
    //   B half: * (-step)            ref: src/poismf.c:523-524
    //   A half: * (-step) twice      ref: src/poismf.c:573-577 (quirk Q1)
    real_t neg_step = -step_size;
    if (bsum_override != nullptr) {
        s->partials_given = false;   // (declared for a half that computes its own sum: not for a later one)
        neg_step = neg_step_override;
        HIP_TRY(pmf_upload(s->d_bsum, bsum_override, s->k * sizeof(real_t), s->stream));
    } else if (prologue) {
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
        if (prologue && !s->padded_fresh[which ? 0 : 1]) {
            const size_t total = dimF * s->k;
            const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, (size_t)s->num_cu * 16);
            if (blocks > 0) hipLaunchKernelGGL((repad_kernel<real_t>), dim3(blocks), dim3(256), 0, s->stream, F, Fp, dimF, (int)s->k, (int)s->ld);
            HIP_TRY(hipGetLastError());
        }
        Fg = Fp;
        ldF = s->ld;
        Mp = which ? s->dAp : s->dBp;
        if (prologue) {
            s->padded_fresh[which ? 0 : 1] = true;   // the copy of F was just re-derived (or was current)
            s->padded_fresh[which ? 1 : 0] = h.row_begin == 0 && h.row_end == h.dimM;
        }
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
    a.P.neg_step2 = neg_step2;
    a.P.maxupd = (int)std::min<size_t>(p->maxupd, 0x7fffffff);
    a.P.limit_step = p->limit_step;
    a.P.max_cg_it = (int)std::max(1.0, std::min(50.0, (double)(real_t)s->k / 2.0));  // ref: src/poismf.c:342
    static const bool no_prune = getenv("POISMF_HIP_NO_LS_PRUNE") != nullptr;  // testing knob: evaluate every line-search trial
    // (w_mult > 0: the bound that lets a line search skip a trial needs the data term -w sum x log(.) to be CONVEX along the line;
    // the reference's Python wrapper asserts weight_mult > 0, the C ABI does not)
    a.P.x_pos = (h.x_positive && !no_prune && p->w_mult > (real_t)0) ? 1 : 0;
    a.reuse_prev = p->reuse_prev;
    a.early_stop = (p->method == POISMF_TNCG) && p->early_stop && (n_unchanged != nullptr || seg >= 0);
    a.n_unchanged = s->d_counter;
    a.eval_rows = s->profiling ? h.d_eval_rows : nullptr;
    a.dec_rows = s->profiling ? h.d_dec_rows : nullptr;
    if (a.early_stop && prologue) HIP_TRY(hipMemsetAsync(s->d_counter, 0, sizeof(unsigned), s->stream));

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
    struct Launch { unsigned begin, count; TileGeom geom; int nw; int reg_S; int team; unsigned long long nnz; int lane_L = 0, lane_A = 0, lane_LL = 0, lane_small = 0, lane_LP = 0, lane_tx = 0; };
    std::vector<Launch> launches;
    // (every team launch of a call gets a row-queue head, an error word and a buffer area of its own: no more than TEAM_LAUNCH_MAX of them; a bin
    // that would open one more takes the path it has without teams)
    auto n_team_launches = [&]() { int n = 0; for (const Launch& L : launches) n += L.team > 1; return n; };
    static const bool no_reg = getenv("POISMF_HIP_NO_REGTILE") != nullptr;  // testing knob: LDS engine for every row
    // register engine: factor rows of at most 16 slots (32 for doubles, two slots per lane), and 24-bit row ids / 32-bit
    // byte offsets into the factor
    const int reg_ns = reg_slots_per_lane((s->k * sizeof(real_t) + 15) / 16);
    const bool reg_ok = !no_reg && reg_ns > 0 &&
                        dimF < ((size_t)1 << 24) && (dimF + 1) * ldF * sizeof(real_t) + 16 < ((size_t)1 << 32);
    // two slots per lane: single-wave rows only, and TNC's ~21 k-vectors leave room for 112 nonzeros of tile
    const bool regw_ok = reg_ok && (reg_ns == 1 || REG_G == 8);
    // (kernel-resource-usage: CG with 40 steps of two slots spills 360 bytes per lane even at one wave per SIMD, 36 steps 60)
    const unsigned reg_max = reg_ns == 2 && REG_G == 16 ? (pm == POISMF_TNCG ? 112u : pm == POISMF_CG ? 144u : reg_nnz_max(pm))
                                                        : reg_nnz_max(pm);
    // teams: CG on doubles with two slots per lane (k = 50 fp64: 25 slots), rows handed out through the queue
    static const bool no_team_env = getenv("POISMF_HIP_NO_TEAM") != nullptr;  // testing knob
    const bool no_team = no_team_env || s->teams_off;
    static const bool static_rows_ = getenv("POISMF_HIP_STATIC_ROWS") != nullptr;
    const bool team_ok = !no_team && !static_rows_ && reg_ok && reg_ns == 2 && REG_G == 16 && sizeof(real_t) == 8 && pm == POISMF_CG;
    // lane-per-nonzero engine (lane_eval.hpp): doubles with 25 / 50 slots per factor row, CG and TNCG; 24-bit row ids and
    // row strides, 32-bit byte offsets into the factor (as the register engine)
    static const bool no_lane = getenv("POISMF_HIP_NO_LANE") != nullptr;  // testing knob
    const bool lane_ok = !no_lane && !single_pass && dimF < ((size_t)1 << 24) && ldF * sizeof(real_t) < ((size_t)1 << 24) &&
                         (dimF + 1) * ldF * sizeof(real_t) + 16 < ((size_t)1 << 32);
    unsigned long_thr = LONG_ROW_NNZ;
    if (const char* e = getenv("POISMF_HIP_LONGROW_NNZ")) long_thr = (unsigned)std::max(64, atoi(e));  // testing knob (a huge value: no eight-wave rows at all)
    const bool no_long = long_thr >= 0x40000000u;
    std::vector<Bin> bins;   // of the segments this call runs, in order
    for (size_t j = 0; j < h.segs.size(); j++)
        if (seg < 0 || (size_t)seg == j) bins.insert(bins.end(), h.segs[j].bins.begin(), h.segs[j].bins.end());
    for (const Bin& b : bins) {
        TileGeom g = plan_geom(s->k, b.cls, single_pass, pm == POISMF_CG && p->limit_step);
        if (single_pass) { g.resident = 0; g.prefetch = prefetch_enabled() ? 1 : 0; }  // one pass: "gather once" and "stream" are the same thing
        if (lane_ok) {
            LaneShape ls = lane_shape_for(b.cls, g.s_load, pm);
            // k = 100 fp64 rows above 64 nonzeros on the B half: rounds 3-4 left them to the streamed launch (with only the 65 .. 128-nonzero rows
            // taken out, that launch lost the short-row tail that kept its wave slots busy: B half 234.6 -> 296.0 ms); since round 5 every row up
            // to 384 nonzeros has a resident instance and the streamed launch keeps the 3 k rows above.
            // k = 100 fp64 under TNCG, rows of 385 .. 8192 nonzeros (round 5): a TEAM of ceil(class / 384) four-wave workgroups keeps the row
            // RESIDENT (each member its 1/M of the nonzeros in one register set + a partial LDS set per wave, lane_eval.hpp TM_) and the members
            // exchange their sums per evaluation -- instead of re-streaming 800 bytes per nonzero for each of ~70 evaluations (84 % of config C5's
            // 697 GB per sweep).  The team size is a function of the row's length class alone.  POISMF_HIP_NO_LANE_TEAMS=1: the eight-wave
            // streamed kernel (round 5a)
            int lane_team = 0;
            static const bool no_lane_teams = getenv("POISMF_HIP_NO_LANE_TEAMS") != nullptr;
            if (ls.waves == 0 && sizeof(real_t) == 8 && g.s_load == 50 && pm == POISMF_TNCG && !no_lane_teams && !no_team && !static_rows_ && b.cls > 384 &&
                b.cls <= LONG_ROW_NNZ && n_team_launches() < TEAM_LAUNCH_MAX - 1) {
                const int m = (int)((b.cls + 383u) / 384u);
                if (m >= 2 && s->num_cu >= 2 * m) { ls = LaneShape{ 1, 0, 0, 4, 0, 32 }; lane_team = m; }
            }
            if (ls.waves > 0) {
                if (!launches.empty() && launches.back().team == lane_team && launches.back().lane_L == ls.lv && launches.back().lane_A == ls.la && launches.back().lane_LL == ls.ll && launches.back().lane_small == ls.small && launches.back().lane_LP == ls.lp && launches.back().lane_tx == ls.tx &&
                    launches.back().nw == ls.waves && launches.back().begin + launches.back().count == b.begin)
                    { launches.back().count += b.count; launches.back().nnz += b.nnz; }
                else {
                    launches.push_back({ b.begin, b.count, g, ls.waves, 0, lane_team, b.nnz });
                    launches.back().lane_L = ls.lv; launches.back().lane_A = ls.la; launches.back().lane_LL = ls.ll; launches.back().lane_small = ls.small; launches.back().lane_LP = ls.lp; launches.back().lane_tx = ls.tx;
                }
                continue;
            }
        }
        if (reg_ok && b.cls <= reg_max) {
            // short rows: the tile lives in registers (reg_eval.hpp); bins sharing a step count share a launch
            // (a bin of a few thousand rows is not worth a launch of its own: it rides along with the next longer size)
            // (TNC keeps the tile size its length class names: in fp32 its results move in the last bits with the size of
            // the instance -- 62 of 900 rows in tests/test_gpu_parity.py's segment test -- and a row must not depend on
            // which other rows share its shard; PG and CG are bit-identical across instances and may ride along)
            const bool ride = pm != POISMF_TNCG || sizeof(real_t) == 8;   // (fp64 TNC is bit-identical across instances too)
            const int S = reg_steps_for(ride ? b.max_nnz : b.cls);
            if (!launches.empty() && launches.back().lane_L == 0 && launches.back().nw == 1 && launches.back().reg_S >= S &&
                (launches.back().reg_S == S || (ride && b.count < 4096u)) && launches.back().begin + launches.back().count == b.begin)
                { launches.back().count += b.count; launches.back().nnz += b.nnz; }
            else
                launches.push_back({ b.begin, b.count, g, 1, S, 0, b.nnz });
            continue;
        }
        if (regw_ok && b.cls <= regw_nnz_max(pm)) {
            // medium rows: 2, 4 or 8 waves share a row, each keeps its part of the tile in registers
            const int nw = regw_waves_for(b.cls, pm);
            const bool ride = pm != POISMF_TNCG || sizeof(real_t) == 8;   // (fp64 TNC is bit-identical across instances too)
            const int S = regw_steps_for(ride ? b.max_nnz : b.cls, nw);
            if (!launches.empty() && launches.back().lane_L == 0 && launches.back().nw == nw && launches.back().reg_S >= S &&
                (launches.back().reg_S == S || (ride && b.count < 2048u)) && launches.back().begin + launches.back().count == b.begin)
                { launches.back().count += b.count; launches.back().nnz += b.nnz; }
            else
                launches.push_back({ b.begin, b.count, g, nw, S, 0, b.nnz });
            continue;
        }
        if (team_ok) {
            // rows whose tile fits the registers of two to four CUs, not of one: a team per row (reg_eval.hpp, M_ > 1)
            const TeamShape ts = n_team_launches() < TEAM_LAUNCH_MAX - 1 ? team_shape_for(b.cls) : TeamShape{};   // by the class bound, never by the longest row that happens to be in the bin: a row's
            // share of the tile (my_share: C = ceil(nnz / (NW M))) -- and with it its summation order -- must not depend on its shard
            if (ts.members > 0) {
                if (!launches.empty() && launches.back().team == ts.members && launches.back().reg_S == ts.steps &&
                    launches.back().begin + launches.back().count == b.begin)
                    { launches.back().count += b.count; launches.back().nnz += b.nnz; }
                else
                    launches.push_back({ b.begin, b.count, g, TEAM_NW, ts.steps, ts.members, b.nnz });
                continue;
            }
        }
        // TNCG streams a non-resident row once per evaluation (~70 of them): one wave keeps ~8 KB of gathers in flight (~4 GB/s), and once
        // the lane engine holds every row up to 384 nonzeros the few thousand longer ones are a tail, not a crowd -- config C5, rows of
        // 385 .. 8192 nonzeros on one wave each: 229 ms; on eight-wave workgroups: inside the 133 ms of the then-giant-row-bound launch.
        // So TNCG's streamed rows always take the eight-wave kernel (POISMF_HIP_LONGROW_NNZ overrides; CG caches its line search, PG
        // makes one gather per pass over the whole chip: they keep the one-wave streamed kernel below 8192 nonzeros).
        const unsigned long_thr_here = (pm == POISMF_TNCG && !g.resident && getenv("POISMF_HIP_LONGROW_NNZ") == nullptr) ? 0u : long_thr;
        // a workgroup of LONG_NW waves per row; every wave streams its own chunks: size the chunk so that
        // LONG_NW private tiles and the reduction scratch fit in one CU's LDS
        TileGeom gl = g;
        gl.resident = 0;
        gl.prefetch = prefetch_enabled() ? 1 : 0;
        gl.pq_cap = 0;
        for (int cap = 128;; cap -= 16) {
            gl.cap = cap;
            if (cap <= 16 || lds_bytes_per_block(gl, sizeof(real_t), LONG_NW) <= 150 * 1024) break;
        }
        // (round 6: eight private tiles of even 16 nonzeros do not fit a CU's LDS once a factor row is ~1.2 KB -- k > 146 in fp64, > 292 in fp32 --
        // and the launch failed with "invalid argument", i.e. rc 1 for a TNCG fit at k = 200 fp64 with any row past the resident limit, found by
        // scripts/knob_matrix.sh under POISMF_HIP_LONGROW_NNZ=256: such rows keep the one-wave streamed kernel below)
        const bool long_fits = lds_bytes_per_block(gl, sizeof(real_t), LONG_NW) <= LDS_PER_CU;
        if (!no_long && long_fits && b.cls > long_thr_here) {
            g = gl;
            // TNCG re-streams such a row for every evaluation: a team of GT_M workgroups per row (row_eval.hpp, TM; POISMF_HIP_NO_GIANT_TEAMS=1:
            // one workgroup per row, rounds 1-4).  Decided by the solver alone: a row's arithmetic must not depend on its shard.
            static const bool no_giant = getenv("POISMF_HIP_NO_GIANT_TEAMS") != nullptr;
            static const unsigned giant_thr = getenv("POISMF_HIP_GIANT_NNZ") ? (unsigned)std::max(64, atoi(getenv("POISMF_HIP_GIANT_NNZ"))) : LONG_ROW_NNZ;   // testing knob
            const int giant = (!no_giant && !no_team && !static_rows_ && pm == POISMF_TNCG && b.cls > giant_thr && n_team_launches() < TEAM_LAUNCH_MAX - 1 &&
                               s->num_cu >= 2 * GT_M) ? GT_M : 0;
            if (!launches.empty() && launches.back().lane_L == 0 && launches.back().reg_S == 0 && launches.back().nw == LONG_NW && launches.back().team == giant &&
                launches.back().begin + launches.back().count == b.begin)
                { launches.back().count += b.count; launches.back().nnz += b.nnz; }
            else
                launches.push_back({ b.begin, b.count, g, LONG_NW, 0, giant, b.nnz });
            continue;
        }
        if (!launches.empty() && launches.back().lane_L == 0 && launches.back().reg_S == 0 && launches.back().geom.cap == g.cap &&
            launches.back().geom.resident == g.resident && (g.resident == 0) && g.pq_cap == 0 && launches.back().geom.pq_cap == 0 && launches.back().nw == 1 &&
            launches.back().begin + launches.back().count == b.begin)
            { launches.back().count += b.count; launches.back().nnz += b.nnz; }
        else
            launches.push_back({ b.begin, b.count, g, 1, 0, 0, b.nnz });
    }
    static const bool static_rows = getenv("POISMF_HIP_STATIC_ROWS") != nullptr;  // testing knob
    const bool dynamic = !is_pg && !static_rows && launches.size() <= (size_t)MAX_LAUNCHES;
    // (PG's multi-wave lane launches take ONE ROW PER WORKGROUP, below; persistent workgroups on the queue or with static shares -- rounds 2-4a,
    // POISMF_HIP_PG_LANE_ROWS -- lost to it, DESIGN.md 6.0, and went in round 6)
    if (dynamic) HIP_TRY(hipMemsetAsync(s->d_queue, 0, sizeof(unsigned) * MAX_LAUNCHES, s->stream));
    // the few workgroup-per-row launches of the power-law tail occupy a few dozen CUs for a long time: run them on a
    // second stream beside the other bins (fork after the column sums, join before anything reads the result)
    static const bool no_fork = getenv("POISMF_HIP_NO_FORK") != nullptr;  // testing knob
    bool any_long = false;
    for (const Launch& L : launches) any_long = any_long || (L.nw > 1 && L.reg_S == 0 && L.lane_L == 0);
    const bool forked = !no_fork && launches.size() > 1 && any_long;
    // (two TEAM launches must never run beside each other: each waits for partners that need the CUs the other's partial teams hold -- giant rows
    // and lane teams follow one another on the SECOND stream, beside the main stream's non-team bins)
    hipStream_t long_stream = forked ? s->aux_stream : s->stream;
    double queued[2] = { 0.0, 0.0 };
    // The long rows go to the second stream to run NEXT TO the other bins, not after them.  The other bins' kernels are persistent
    // (a workgroup keeps its CU until the bin's queue is empty): whichever kernel reaches the chip first fills it, and on config C5
    // that was the mid-length bin -- the 60 giant rows then waited 260 ms for a CU and ran on their own afterwards (390 ms for what
    // takes 150 alone).  So every workgroup of the long-row launch counts itself in when it starts, and the main stream waits for
    // that count (a one-wave gate kernel with a time limit, above; rounds 3-4a: hipStreamWaitValue32) before it launches anything else.
    // Arrivals only ever grow, so a chip that cannot hold the whole launch at once delays the main stream by the gate's 2 ms, no more.
    const bool hold_back = forked && any_long;
    if (hold_back) HIP_TRY(hipMemsetAsync(s->d_arrive, 0, sizeof(unsigned), s->stream));
    // ---- everything the team launches of this call need, once, on the main stream, before anything of the half is on the chip: a zeroed
    // buffer area, row-queue head (+ one for the re-run), error word and tally per launch, and a copy of the rows they start from
    struct TeamSlot { size_t backup_at, eval_at; int area; };   // per team launch, in launch order
    std::vector<TeamSlot> tslots;
    {
        size_t elems = 0, rows = 0;
        int n_gt = 0, n_reg = 0;
        for (const Launch& L : launches) {
            if (L.team <= 1) continue;
            const bool gt = L.lane_L > 0 || (L.team == GT_M && L.reg_S == 0);   // lane teams and giant rows share the GT_* layout
            tslots.push_back({ elems, rows, gt ? n_gt++ : n_reg++ });
            elems += (size_t)L.count * s->k;
            rows += L.count;
        }
        if (!tslots.empty()) {
            // (growing a buffer frees it first, i.e. synchronises the device: sized for the whole half at once, and only ever grown)
            if ((size_t)n_gt > s->gt_areas) {
                pmf_free(s->d_gt, s->stream); s->d_gt = nullptr; s->gt_areas = 0;
                HIP_TRY(pmf_alloc(&s->d_gt, (size_t)n_gt * (size_t)GT_BUF_BYTES, s->stream));
                s->gt_areas = (size_t)n_gt;
            }
            if ((size_t)n_reg > s->team_areas) {
                pmf_free(s->d_team, s->stream); s->d_team = nullptr; s->team_areas = 0;
                HIP_TRY(pmf_alloc(&s->d_team, (size_t)n_reg * (size_t)TEAM_BUF_BYTES, s->stream));
                s->team_areas = (size_t)n_reg;
            }
            if (elems > s->team_backup_elems) {
                pmf_free(s->d_team_backup, s->stream); s->d_team_backup = nullptr; s->team_backup_elems = 0;
                HIP_TRY(pmf_alloc(&s->d_team_backup, elems * sizeof(real_t), s->stream));
                s->team_backup_elems = elems;
            }
            if (a.eval_rows != nullptr && rows > s->team_eval_backup_rows) {
                pmf_free(s->d_team_eval_backup, s->stream); s->d_team_eval_backup = nullptr; s->team_eval_backup_rows = 0;
                HIP_TRY(pmf_alloc(&s->d_team_eval_backup, rows * sizeof(unsigned), s->stream));
                s->team_eval_backup_rows = rows;
            }
            // (a giant / lane team's area is used up to its teams' words: the whole areas are zeroed all the same -- n x 4.6 MB, microseconds on an
            // empty chip, where round 5 zeroed one area per launch between persistent kernels)
            if (n_gt) HIP_TRY(hipMemsetAsync(s->d_gt, 0, (size_t)n_gt * (size_t)GT_BUF_BYTES, s->stream));
            if (n_reg) HIP_TRY(hipMemsetAsync(s->d_team, 0, (size_t)n_reg * (size_t)TEAM_BUF_BYTES, s->stream));
            HIP_TRY(hipMemsetAsync(s->d_queue + MAX_LAUNCHES, 0, sizeof(unsigned) * 2 * TEAM_LAUNCH_MAX, s->stream));
            HIP_TRY(hipMemsetAsync(s->d_team_err + 2, 0, sizeof(unsigned) * 2 * TEAM_LAUNCH_MAX, s->stream));
            size_t ti = 0;
            for (const Launch& L : launches) {
                if (L.team <= 1) continue;
                const size_t need = (size_t)L.count * s->k;
                hipLaunchKernelGGL(team_save_rows_kernel, dim3((unsigned)std::min<size_t>((need + 255) / 256, (size_t)s->num_cu * 8)), dim3(256), 0, s->stream,
                                   M, h.d_desc + L.begin, L.count, (unsigned)h.row_begin, (int)s->k, s->d_team_backup + tslots[ti].backup_at,
                                   (const unsigned*)a.eval_rows, a.eval_rows != nullptr ? s->d_team_eval_backup + tslots[ti].eval_at : nullptr);
                ti++;
            }
            HIP_TRY(hipGetLastError());
            s->team_launched = true;
        }
    }
    struct TeamRerun { HalfArgs<real_t> af; OneLaunch of; unsigned begin, count; size_t slot; };
    std::vector<TeamRerun> reruns;   // what each team launch becomes if it gives up (issued, gated, after the join)
    size_t team_no = 0;
    if (forked) {
        HIP_TRY(hipEventRecord(s->ev_fork, s->stream));
        HIP_TRY(hipStreamWaitEvent(s->aux_stream, s->ev_fork, 0));
    }
    int launch_no = 0;
    unsigned arrive_goal = 0;
    if (prologue) s->last_plan[which].clear();
    for (const Launch& L : launches) {
        char lname[160];
        {
            char txt[192];
            const char* m = is_pg ? "pg" : p->method == POISMF_EVAL ? "eval" : pm == POISMF_CG ? "cg" : "tncg";
            const char* t = sizeof(real_t) == 4 ? "float" : "double";
            if (L.lane_L > 0 && L.team > 1) snprintf(txt, sizeof txt, "half_sweep_lane_team_kernel<%s,%s,KS=%d,V=%d,L=0+%d,NW=%d,M=%d> rows=%u;", t, m, L.geom.s_load, L.lane_L, L.lane_LP, L.nw, L.team, L.count);
            else if (L.lane_L > 0) snprintf(txt, sizeof txt, "half_sweep_lane_kernel<%s,%s,KS=%d,V=%d,A=%d,L=%d%s,NW=%d%s%s> rows=%u;", t, m, L.geom.s_load, L.lane_L, L.lane_A, L.lane_LL, L.lane_LP == 32 ? "+32" : L.lane_LP ? "+16" : "", L.nw, L.lane_small ? ",2/SIMD" : "", L.lane_tx == 48 ? ",TX=48" : L.lane_tx == 64 ? ",TX=64" : "", L.count);
            else if (L.team == GT_M && L.reg_S == 0) snprintf(txt, sizeof txt, "half_sweep_giant_kernel<%s,%s,NW=%d,M=%d,streamed cap=%d> rows=%u;", t, m, L.nw, L.team, L.geom.cap, L.count);
            else if (L.team > 1) snprintf(txt, sizeof txt, "half_sweep_team_kernel<%s,%s,S=%d,NW=%d,M=%d> rows=%u;", t, m, L.reg_S, L.nw, L.team, L.count);
            else if (L.reg_S > 0 && L.nw == 1) snprintf(txt, sizeof txt, "half_sweep_reg_kernel<%s,%s,S=%d> rows=%u;", t, m, L.reg_S, L.count);
            else if (L.reg_S > 0) snprintf(txt, sizeof txt, "half_sweep_regw_kernel<%s,%s,S=%d,NW=%d> rows=%u;", t, m, L.reg_S, L.nw, L.count);
            else snprintf(txt, sizeof txt, "half_sweep_kernel<%s,%s,NW=%d,%s cap=%d> rows=%u;", t, m, L.nw, L.geom.resident ? "resident" : "streamed", L.geom.cap, L.count);
            s->last_plan[which] += txt;
            snprintf(lname, sizeof lname, "%s", txt);
            if (char* sp = strstr(lname, " rows=")) *sp = 0;
        }
        a.queue = dynamic ? s->d_queue + launch_no : nullptr;
        a.stop = is_pg ? nullptr : (const unsigned*)g_stop_word;   // (pinned host memory, portable: the same address on every device)
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
        const bool one_wave_reg = (L.reg_S > 0 || L.lane_L > 0) && L.nw == 1;
        if (one_wave_reg) a.queue = nullptr;
        a.team_buf = nullptr; a.team_err = s->d_team_err + 2;
        a.n_unchanged = s->d_counter;
        unsigned* terr = s->d_team_err + 2;
        const bool giant = L.team == GT_M && L.reg_S == 0 && L.lane_L == 0;
        const bool lane_team = L.team > 1 && L.lane_L > 0;
        a.team_members = (unsigned)std::max(1, L.team);
        // (a giant-row team launch lives on the stream the long rows run on)
        hipStream_t tst = (giant || lane_team) ? long_stream : s->stream;
        const size_t my_slot = team_no;
        if (L.team > 1) {
            // one dispatch: its queue head, buffer area, error word and tally are its own and were zeroed before the half began
            team_no++;
            a.queue = s->d_queue + MAX_LAUNCHES + my_slot;            // teams always draw their rows from a queue
            terr = s->d_team_err + 2 + my_slot;
            a.team_err = terr;
            a.n_unchanged = s->d_team_err + 2 + TEAM_LAUNCH_MAX + my_slot;   // kept only if the launch's results are (team_fold_kernel)
            a.team_buf = (giant || lane_team) ? s->d_gt + (size_t)tslots[my_slot].area * (size_t)(GT_BUF_BYTES / 8)
                                              : s->d_team + (size_t)tslots[my_slot].area * (size_t)(TEAM_BUF_BYTES / 8);
        }
        a.gate = nullptr;
        const bool is_long = L.nw > 1 && L.reg_S == 0 && L.lane_L == 0;
        a.arrive = hold_back && is_long ? s->d_arrive : nullptr;
        static const unsigned team_spin = getenv("POISMF_HIP_TEAM_SPIN_LIMIT") ? (unsigned)std::max(1, atoi(getenv("POISMF_HIP_TEAM_SPIN_LIMIT"))) : TEAM_SPIN_LIMIT;   // testing knob
        a.team_spin = team_spin;
        unsigned grid_mult = one_wave_reg ? 32 : 2;
        // PG on the multi-wave lane kernel: ONE ROW PER WORKGROUP, the hardware dispatcher hands them out.  C4 matrix, PG(10), the 78 715 item
        // rows of 513 .. 1024 nonzeros: persistent workgroups walking rows r, r + grid, .. at 2 / 4 / 8 / 16 / 64 workgroups per slot 4.28 /
        // 4.13 / 4.08 / 4.09 / 4.39 ms; persistent workgroups on the queue 4.07; one row per workgroup 3.87 ms.  The queue's gain is balance
        // (no workgroup owns a fixed share of the rows); what the dispatcher gains on top is measured, not explained (DESIGN.md section 6.0:
        // neither the cross-row pipeline nor start delays account for it; the workgroup-wide ticket's two barriers per row remain).
        // (Not for CG / TNCG, whose rows differ in cost and want the longest-first queue: CG fp32 B half 11.25 -> 13.17 ms; not for the
        // eight-wave register kernel, one workgroup per CU: 1.83 -> 1.90.)
        if (is_pg && L.lane_L > 0 && L.nw > 1) grid_mult = 1u << 20;
        unsigned grid = (unsigned)std::min<size_t>(L.count, (size_t)s->num_cu * waves_per_cu * grid_mult);
        if (giant) {
            // whole teams only: as many as the chip holds at one workgroup per CU
            const unsigned teams = std::max(1u, std::min((unsigned)L.count, std::min((unsigned)GT_TEAMS_MAX, (unsigned)s->num_cu / (unsigned)GT_M)));
            grid = teams * (unsigned)GT_M;
        }
        if (lane_team) grid = std::max(1u, std::min((unsigned)L.count, (unsigned)s->num_cu / (unsigned)L.team)) * (unsigned)L.team;   // whole teams, one workgroup per CU
        int rc = 1;
        // (with long rows on the second stream, the one-wave bins that follow go wherever less work is queued -- unless the half has TEAM launches:
        // they follow one another on the second stream and are the half's critical path; round 6's timeline of a C5 sweep, profiles/r06/kt_c5_timeline.txt,
        // showed the last one-wave bin queued behind all seven of them and running alone for 8 ms after the main stream had been idle for 33)
        const int lane_stream = (forked && tslots.empty() && L.nw == 1 && queued[1] < queued[0]) ? 1 : 0;
        hipStream_t bin_stream = lane_stream ? s->aux_stream : s->stream;
        queued[L.nw > 1 ? 1 : lane_stream] += (double)L.count * (double)std::max(16, L.reg_S > 0 ? L.reg_S * REG_JG : L.geom.cap);
        {
            OneLaunch o;
            o.reg_S = L.reg_S; o.nw = L.nw; o.team = L.team; o.lane_L = L.lane_L; o.lane_A = L.lane_A; o.lane_LL = L.lane_LL; o.lane_small = L.lane_small; o.lane_LP = L.lane_LP; o.lane_tx = L.lane_tx; o.s_load = a.geom.s_load; o.spl = slots_per_lane(s->k);
            o.generic_only = false;   // (the generic slot-count kernels are what every k other than the BASELINE configs' takes: tests/test_gpu_regtile.py)
            o.main_stream = lane_team ? tst : s->stream; o.bin_stream = bin_stream; o.long_stream = long_stream;
            o.lds = lds; o.grid = grid; o.grid_mult = grid_mult;
            o.device = s->device; o.num_cu = s->num_cu;
            // profiling sessions: events around this launch, on the stream it goes to (launch_one_here's choice)
            hipStream_t lst = giant ? o.long_stream : L.team > 1 || ((L.reg_S > 0 || L.lane_L > 0) && L.nw > 1) ? o.main_stream : (L.reg_S == 0 && L.lane_L == 0 && L.nw > 1 ? o.long_stream : o.bin_stream);
            LaunchRec lr{};
            if (s->profiling) {
                HIP_TRY(hipEventCreate(&lr.t0));
                HIP_TRY(hipEventCreate(&lr.t1));
                lr.which = which; lr.name = lname; lr.rows = L.count; lr.nnz = L.nnz;
                HIP_TRY(hipEventRecord(lr.t0, lst));
            }
            rc = launch_one(p->method, o, a);
            if (!rc && a.arrive != nullptr) {
                arrive_goal += std::min<unsigned>(grid, (unsigned)s->num_cu);   // (one eight-wave workgroup per CU)
                // (hold_back_gate_kernel: returns when that many workgroups are on the chip, or after 2 ms)
                hipLaunchKernelGGL(hold_back_gate_kernel, dim3(1), dim3(1), 0, s->stream, s->d_arrive, arrive_goal, s->gate_budget);
            }
            if (!rc && L.team > 1) {
                // if the team launch gives up: rows back to where they started, the same rows on the streamed LDS kernel -- after the join (below)
                HalfArgs<real_t> af = a;
                // (a.geom is the LDS engine's geometry for the launch's longest length class: what these rows take without teams)
                af.team_buf = nullptr;
                af.gate = terr;
                af.arrive = nullptr;
                af.n_unchanged = s->d_counter;
                af.queue = s->d_queue + MAX_LAUNCHES + TEAM_LAUNCH_MAX + my_slot;
                if (lane_team) {   // (a lane launch carries the one-wave LDS geometry of its class: the eight-wave streamed kernel wants its own)
                    af.geom.resident = 0;
                    af.geom.prefetch = prefetch_enabled() ? 1 : 0;
                    af.geom.pq_cap = 0;
                    for (int cap = 128;; cap -= 16) {
                        af.geom.cap = cap;
                        if (cap <= 16 || lds_bytes_per_block(af.geom, sizeof(real_t), LONG_NW) <= 150 * 1024) break;
                    }
                }
                OneLaunch of = o;
                of.reg_S = 0; of.nw = (giant || lane_team) ? LONG_NW : 1; of.team = 0; of.lane_L = 0; of.lane_A = 0; of.lane_LL = 0; of.lane_small = 0; of.lane_LP = 0; of.lane_tx = 0;
                of.s_load = af.geom.s_load;
                of.main_stream = s->stream; of.bin_stream = s->stream; of.long_stream = s->stream;
                of.lds = lds_bytes_per_block(af.geom, sizeof(real_t), of.nw);
                of.grid = (giant || lane_team) ? (unsigned)std::min<size_t>(L.count, (size_t)s->num_cu)
                                : (unsigned)std::min<size_t>(L.count, (size_t)s->num_cu * std::max<size_t>(1, std::min<size_t>(16, LDS_PER_CU / of.lds)) * 2);
                reruns.push_back({ af, of, L.begin, L.count, my_slot });
            }
            if (s->profiling) {
                HIP_TRY(hipEventRecord(lr.t1, lst));
                s->lprof.push_back(lr);
            }
        }
        if (rc) return 1;
    }
    if (forked) {
        HIP_TRY(hipEventRecord(s->ev_join, s->aux_stream));
        HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_join, 0));
    }
    // the team launches' epilogue, on a chip that has nothing else resident: per launch a restore and a streamed re-run that return at once unless
    // the launch's error word is set, then the fold
    for (const TeamRerun& r : reruns) {
        hipLaunchKernelGGL(team_restore_rows_kernel, dim3((unsigned)std::min<size_t>(((size_t)r.count * s->k + 255) / 256, (size_t)s->num_cu * 8)),
                           dim3(256), 0, s->stream, M, Mp, (int)s->ld, h.d_desc + r.begin, r.count, (unsigned)h.row_begin, (int)s->k,
                           s->d_team_backup + tslots[r.slot].backup_at, s->d_team_err + 2 + r.slot, a.eval_rows,
                           a.eval_rows != nullptr ? s->d_team_eval_backup + tslots[r.slot].eval_at : nullptr);
#ifndef PMF_LANE_ONLY   // (development builds without the streamed kernels: no re-run)
        if (launch_one(p->method, r.of, r.af)) return 1;
#endif
    }
    if (!reruns.empty()) {
        hipLaunchKernelGGL(team_fold_kernel, dim3(1), dim3(1), 0, s->stream, s->d_team_err, (int)tslots.size(), a.early_stop ? s->d_counter : nullptr);
        HIP_TRY(hipGetLastError());
    }
    if (s->profiling) {
        HIP_TRY(hipEventRecord(rec.t1, s->stream));
        s->prof.push_back(rec);
    }
    if (a.early_stop && n_unchanged != nullptr) {
        unsigned cnt = 0;
        HIP_TRY(pmf_download(&cnt, s->d_counter, sizeof(unsigned), s->stream));
        *n_unchanged = cnt;
    }
    return 0;
}

int poismf_hip_half_sweep(poismf_hip_session* s, int which, const poismf_hip_params* p, real_t step_size, real_t cnst_div,
                          size_t* n_unchanged)
{
    return half_sweep_impl(s, which, p, step_size, cnst_div, n_unchanged, nullptr, (real_t)0);
}

// The launches of the most recent half-sweep of half `which` ("kernel<instance> rows=N;" per launch), NUL-terminated,
// truncated to cap bytes.  Returns the untruncated length.
size_t poismf_hip_session_plan(poismf_hip_session* s, int which, char* buf, size_t cap)
{
    const std::string& t = s->last_plan[which ? 1 : 0];
    if (cap > 0) {
        const size_t n = std::min(cap - 1, t.size());
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return t.size();
}

// Per-launch durations of half `which` since profile(1), launches of the same instance and row count added up:
// "kernel<instance> rows=R nnz=Z calls=C ms=T;" per distinct launch (T = summed milliseconds).  NUL-terminated, truncated to
// cap bytes; returns the untruncated length.
size_t poismf_hip_session_launch_profile(poismf_hip_session* s, int which, char* buf, size_t cap)
{
    (void)hipStreamSynchronize(s->stream);
    (void)hipStreamSynchronize(s->aux_stream);
    struct Agg { std::string name; unsigned rows; unsigned long long nnz; unsigned calls; double ms; };
    std::vector<Agg> agg;
    for (auto& r : s->lprof) {
        if (r.which != (which ? 1 : 0)) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.t0, r.t1) != hipSuccess) continue;
        bool found = false;
        for (auto& g : agg)
            if (g.name == r.name && g.rows == r.rows && g.nnz == r.nnz) { g.calls++; g.ms += ms; found = true; break; }
        if (!found) agg.push_back({ r.name, r.rows, r.nnz, 1u, (double)ms });
    }
    std::string t;
    for (auto& g : agg) {
        char txt[256];
        snprintf(txt, sizeof txt, "%s rows=%u nnz=%llu calls=%u ms=%.6f;", g.name.c_str(), g.rows, g.nnz, g.calls, g.ms);
        t += txt;
    }
    if (cap > 0) {
        const size_t n = std::min(cap - 1, t.size());
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return t.size();
}

// Serving from the session's resident factors: predict_multiple (ref: src/pred.c:42-64) and topN for the
// user in row `user` of A (ref: src/topN.c:112-284) without copying the factors per call.
int poismf_hip_session_predict(poismf_hip_session* s, const sparse_ix* ixA, const sparse_ix* ixB, size_t n, real_t* out)
{
    if (n == 0) return 0;
    for (size_t i = 0; i < n; i++)
        if ((size_t)ixA[i] >= s->dimA || (size_t)ixB[i] >= s->dimB) return 2;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return poismf_hip_serve_predict(s->dA, s->dB, ixA, ixB, n, (int)s->k, out, nullptr, nullptr);
}

int poismf_hip_session_topn(poismf_hip_session* s, size_t user, const sparse_ix* include_ix, size_t n_include, const sparse_ix* exclude_ix,
                            size_t n_exclude, sparse_ix* outp_ix, real_t* outp_score, size_t n_top)
{
    if (user >= s->dimA) return 2;
    if (const int rc = poismf_hip_serve_topn_check(include_ix, n_include, exclude_ix, n_exclude, n_top, s->dimB)) return rc;
    for (size_t i = 0; include_ix && i < n_include; i++)
        if ((size_t)include_ix[i] >= s->dimB) return 2;
    for (size_t i = 0; exclude_ix && i < n_exclude; i++)
        if ((size_t)exclude_ix[i] >= s->dimB) return 2;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return poismf_hip_serve_topn(s->dA + user * s->k, s->dB, (int)s->k, include_ix, n_include, exclude_ix, n_exclude, outp_ix, outp_score,
                                 n_top, s->dimB);
}

#ifdef PMF_PROBE
// development only: the head words of the team buffer (a -DPMF_PROBE build sums phase cycles of the last team launch in [8, 16))
extern "C" __attribute__((visibility("default"))) int poismf_hip_debug_team_head(poismf_hip_session* s, unsigned long long* out)
{
    if (s->d_team == nullptr) return 1;
    return pmf_download(out, s->d_team, 8 * TEAM_HEAD_WORDS, s->stream) != hipSuccess;
}
// development only (not in the header): the raw per-row counters of half `which`, which a -DPMF_PROBE build fills with stamps
extern "C" __attribute__((visibility("default"))) int poismf_hip_debug_eval_rows(poismf_hip_session* s, int which, unsigned* out, size_t n)
{
    Half& h = s->half[which ? 1 : 0];
    if (h.d_eval_rows == nullptr) return 1;
    return pmf_download(out, h.d_eval_rows, sizeof(unsigned) * std::min(n, h.row_end - h.row_begin), s->stream) != hipSuccess;
}
#endif

int poismf_hip_session_set_segments(poismf_hip_session* s, int which, int nseg)
{
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    Half& h = s->half[which ? 1 : 0];
    if (h.d_indptr == nullptr || finish_half(h, s->stream, nseg)) return -1;
    return (int)h.segs.size();
}

int poismf_hip_session_segment_rows(poismf_hip_session* s, int which, int seg, size_t* row_begin, size_t* row_end)
{
    const Half& h = s->half[which ? 1 : 0];
    if (seg < 0 || seg >= (int)h.segs.size()) return 1;
    *row_begin = h.row_begin + h.segs[seg].row_lo;
    *row_end = h.row_begin + h.segs[seg].row_hi;
    return 0;
}

int poismf_hip_half_sweep_segment(poismf_hip_session* s, int which, const poismf_hip_params* p, real_t step_size, real_t cnst_div,
                                  int seg, size_t* n_unchanged)
{
    return half_sweep_impl(s, which, p, step_size, cnst_div, n_unchanged, nullptr, (real_t)0, (real_t)1, seg);
}

// -------------------------------------------------------------------------------------------------
// run_poismf: the drop-in                                   ref: src/poismf.c:435-632
// -------------------------------------------------------------------------------------------------
static volatile sig_atomic_t g_should_stop = 0;
static bool g_handle_locked = false;
static std::mutex g_handle_mutex;
// The interrupt reaches the row loops of CG / TNCG within a row (ref: src/poismf.c:301, :360: the reference's row loops test
// should_stop_procedure before every row and skip the rest).  Rounds 1-4 looked at the flag between half-sweeps only: up to one half (config
// C5: 260 ms) of latency.  The flag now has a twin in PINNED HOST MEMORY that the device reads directly (HalfArgs::stop, system-scope loads
// next to every row ticket): the handler's own store is all it takes -- no thread, no copy, no kernel that would have to find a free CU
// behind the very workgroups it is meant to stop (a first version overwrote the device-side queue heads by hipMemsetAsync / hipMemcpyAsync:
// 14 .. 290 ms, depending on what the chip was running).  PG has no poll (neither has the reference's pg_iteration, quirk Q8).
static void on_sigint(int)
{
    g_should_stop = 1;  // the reference also prints here; fprintf is not async-signal-safe, so run_poismf reports it
    volatile unsigned* w = g_stop_word;
    if (w != nullptr) *w = 1u;
}
// (g_handle_mutex held, or single-threaded) the device-visible twin of the flag; nullptr if it cannot be had (the flag is then polled
// between half-sweeps only, as before)
static const unsigned* stop_word_for_device()
{
    static bool tried = false;
    static const bool off = getenv("POISMF_HIP_NO_ROW_INTERRUPT") != nullptr;   // testing knob
    if (off) return nullptr;
    if (!tried) {
        tried = true;
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && p != nullptr) {
            *(volatile unsigned*)p = g_should_stop ? 1u : 0u;
            g_stop_word = (volatile unsigned*)p;
        } else (void)hipGetLastError();
    }
    return (const unsigned*)g_stop_word;
}

// SIGINT plumbing of one call (ref: src/poismf.c:444-455, :618-630): the first call in the process to get here installs
// the handler and restores the previous one on the way out; nested / concurrent calls share the flag.
namespace {
struct SigintScope {
    typedef void (*sig_fn)(int);
    sig_fn old_handler = nullptr;
    bool has_lock = false;
    void enter()
    {
        std::lock_guard<std::mutex> lk(g_handle_mutex);
        if (!g_handle_locked) {
            g_handle_locked = true;
            has_lock = true;
            g_should_stop = 0;
            (void)stop_word_for_device();
            if (g_stop_word != nullptr) *g_stop_word = 0u;
            old_handler = signal(SIGINT, on_sigint);
        }
    }
    int leave(int ret_code, bool handle_interrupt)
    {
        std::lock_guard<std::mutex> lk(g_handle_mutex);
        const bool stopped = g_should_stop != 0;
        if (stopped) fprintf(stderr, "Error: procedure was interrupted\n");
        if (stopped && ret_code != 1) ret_code = 2;
        if (has_lock) {
            signal(SIGINT, old_handler);
            g_handle_locked = false;
            g_should_stop = 0;
            if (g_stop_word != nullptr) *g_stop_word = 0u;
        }
        if (stopped && !handle_interrupt) raise(SIGINT);
        return ret_code;
    }
};

// The outer alternation on a session whose factors are set (ref: src/poismf.c:506-608).  Returns 0, or 1 on a device error.
// after_first_b: called once, right after the FIRST B half has been launched (run_poismf uploads the A side's matrix then: that half needs
// the CSC and the factors only, and the copy engine is idle while it runs)
int run_alternation(poismf_hip_session* s, const poismf_hip_params& p, size_t numiter, const std::function<int()>* after_first_b = nullptr)
{
    const int method = p.method;
    const real_t l2_reg = p.l2_reg;
    real_t step_size = p.step_size;
    const bool tn_stop = (method == POISMF_TNCG) && p.early_stop;
    bool stopped_earlyA = false, stopped_earlyB = false;
    for (size_t it = 0; it < numiter; it++) {
        if (g_should_stop) break;
        // quirk Q6: the divisor uses the step before halving and is reused by the A half
        const real_t cnst_div = 1. / (1. + 2. * l2_reg * step_size);

        // ---- B half first (quirk Q5) ----
        if (!(method == POISMF_TNCG && stopped_earlyB)) {
            size_t unchanged = 0;
            if (poismf_hip_half_sweep(s, 0, &p, step_size, cnst_div, tn_stop ? &unchanged : nullptr)) return 1;
            if (tn_stop) stopped_earlyB = ((double)unchanged / (double)s->dimB) >= .95;  // ref: :401-403 (quirk Q7)
        }
        if (it == 0) pmf_tl("first B half launched");
        if (it == 0 && after_first_b != nullptr && (*after_first_b)()) return 1;
        if (method == POISMF_PG) step_size *= 0.5;  // ref: :532-533
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (it == 0) pmf_tl("first B half done");
        if (team_check(s)) return 1;   // (a word of eight bytes, and only after a half that had team launches)
        if (g_should_stop) break;

        // ---- A half ----
        if (!(method == POISMF_TNCG && stopped_earlyA)) {
            size_t unchanged = 0;
            if (poismf_hip_half_sweep(s, 1, &p, step_size, cnst_div, tn_stop ? &unchanged : nullptr)) return 1;
            if (tn_stop) stopped_earlyA = ((double)unchanged / (double)s->dimA) >= .95;
        }
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (it == 0) pmf_tl("first A half done");
        if (team_check(s)) return 1;
        if (stopped_earlyA && stopped_earlyB) break;
    }
    return team_check(s);
}
}  // namespace

// -------------------------------------------------------------------------------------------------
// Several GPUs behind the same C-ABI: POISMF_HIP_DEVICES=0,1,..,7 makes run_poismf cut the rows of A and of B into
// one contiguous, nnz-balanced range per listed device (ref: the row loops it shards are src/poismf.c:159-162, :296-299,
// :352-358), keep one session per device -- its shard of the CSR / CSC, both factors replicated -- and alternate with one host
// thread per device.  After a half every device holds the rows it updated; they travel DIRECTLY to every peer, device to
// device (hipMemcpyPeerAsync on the owner's stream: over xGMI's full mesh all seven links of a GPU carry one shard each at the
// same time; no staging, no collective to wait for the slowest rank), and the next half starts when all of them have landed.
// The k-vector column sums: every device needs the same bits, so the sum is cut into fixed blocks whose partial sums depend on the
// block number alone; device d computes its 1 / D of the blocks over its replica, the [blocks x k] partials travel like the rows, and
// every device runs the fixed-order second stage (round 5; rounds 2-4 had every device recompute the whole sum) -- which is what
// replaces the north-star's all-reduce with sharding-independent bits; TNCG's early-stop counter is summed on the host.
// A device may be listed more than once (POISMF_HIP_DEVICES=0,0): the shards then share that GPU, which is how the one-GPU
// test exercises every line of this path (tests/test_gpu_multi.py); results equal the single-session run bit for bit, since a
// row's arithmetic depends on its length class alone and the column sums are computed in one fixed order.
// -------------------------------------------------------------------------------------------------
extern "C++" {
namespace {

struct Range { size_t lo, hi; };
// contiguous row ranges with (nearly) equal nonzero counts: cuts at the nnz quantiles of the row pointers
std::vector<Range> balanced_ranges(const sparse_ix* indptr, size_t n, size_t parts)
{
    std::vector<Range> out;
    const unsigned long long total = (unsigned long long)indptr[n];
    size_t prev = 0;
    for (size_t pidx = 1; pidx <= parts; pidx++) {
        size_t cut = n;
        if (pidx < parts) {
            const unsigned long long target = total * pidx / parts;
            cut = (size_t)(std::lower_bound(indptr, indptr + n + 1, (sparse_ix)target) - indptr);
            cut = std::min(std::max(cut, prev), n);
        }
        out.push_back({ prev, cut });
        prev = cut;
    }
    return out;
}

// Host threads meet here; the LAST one to arrive runs `last` (decisions every thread must share) before anybody leaves.
struct HostBarrier {
    std::mutex m;
    std::condition_variable cv;
    size_t n, waiting = 0, generation = 0;
    explicit HostBarrier(size_t n_) : n(n_) {}
    template <class Fn> void arrive(Fn&& last)
    {
        std::unique_lock<std::mutex> lk(m);
        const size_t gen = generation;
        if (++waiting == n) {
            last();
            waiting = 0;
            generation++;
            cv.notify_all();
        } else cv.wait(lk, [&] { return generation != gen; });
    }
    void arrive() { arrive([] {}); }
};

// Round 4.  One PERSISTENT host thread per device runs the whole alternation for its device (round 3 created and joined a thread
// per device twice per half and synchronised every stream with the host in between: at C4 on 8 GPUs a PG half is ~0.65 ms per
// device, the same order as those).  Devices are ordered against each other by EVENTS only:
//   * after a half, device d copies the rows it updated straight into every peer's replica (hipMemcpyPeerAsync, xGMI full mesh) on a
//     COPY stream of its own, segment by segment (the A half is cut into segments, poismf_hip_session_set_segments): segment j
//     travels while segment j + 1 computes; the event landed[d] is recorded behind the last copy;
//   * before its next half, device d makes its session stream wait for landed[q] of every peer q.  That one wait covers all three
//     hazards: the rows the next half gathers have arrived; a peer finished READING factor M (as the fixed factor of its previous
//     half) before anybody's copies of M's new rows reach it (those copies follow kernels that waited for that peer's landed event);
//     and d's own copies of two halves ago are done before d overwrites the same rows again (every peer waited for them before
//     the half whose landed event d has just waited for).
//     (TNCG with early stop may skip a half: two CONSECUTIVE halves then update the same factor, and a device may overwrite rows whose
//     previous copies are still travelling.  Harmless: copies on one copy stream are issued and land in order, every reader of those rows waits for
//     the LATER half's landed event, and the replica of a peer ends up holding the later rows -- nobody reads in between.)
// A stream can only wait for an event that has been RECORDED, so the threads hand over "recorded" through an atomic counter per
// device (a host-side spin for the record CALL of a peer, never for the device); two events per device alternate.
// Host threads meet at a barrier once per outer iteration (interrupt flag and failures: everybody takes the same decision) and,
// for TNCG with early stop, once per half (the unchanged-row counts are summed, ref: src/poismf.c:395-403).
struct MultiRun {
    size_t nd;
    const std::vector<int>& devices;
    std::vector<Range> rA, rB;
    std::vector<poismf_hip_session*> ss;
    std::vector<hipStream_t> copy_stream;
    std::vector<hipEvent_t> seg_done;                  // "this segment's kernels are done": session stream -> copy stream
    std::vector<hipEvent_t> landed;                    // [2 d + parity]: device d's rows of a half have reached every peer
    std::unique_ptr<std::atomic<unsigned>[]> recorded; // halves of device d whose `landed` event has been recorded
    std::vector<hipEvent_t> part_done;                 // "this device's share of the column sums' first stage is done": session stream -> copy stream
    std::vector<hipEvent_t> part_landed;               // [2 d + parity]: device d's partial sums of a half have reached every peer
    std::unique_ptr<std::atomic<unsigned>[]> part_recorded;
    std::vector<size_t> unchanged;
    std::vector<hipError_t> err;
    std::atomic<int> failed{0};
    bool stop = false;                                 // decided at the iteration barrier
    HostBarrier bar;
    MultiRun(const std::vector<int>& devs, size_t dimA, size_t dimB, const sparse_ix* pA, const sparse_ix* pB)
        : nd(devs.size()), devices(devs), rA(balanced_ranges(pA, dimA, devs.size())), rB(balanced_ranges(pB, dimB, devs.size())),
          ss(nd, nullptr), copy_stream(nd, nullptr), seg_done(nd, nullptr), landed(2 * nd, nullptr),
          recorded(new std::atomic<unsigned>[nd]), part_done(nd, nullptr), part_landed(2 * nd, nullptr),
          part_recorded(new std::atomic<unsigned>[nd]), unchanged(nd, 0), err(nd, hipSuccess), bar(nd)
    {
        for (size_t d = 0; d < nd; d++) { recorded[d].store(0); part_recorded[d].store(0); }
    }
    void fail(size_t d)
    {
        if (err[d] == hipSuccess) err[d] = pmf_last_hip_error();
        failed.store(1);
    }
    // every peer's rows of half number `h` (0-based) have been copied into THIS device's replica: ordered before whatever is
    // issued next on the session stream
    int wait_for_peers(size_t d, unsigned h)
    {
        for (size_t q = 0; q < nd; q++) {
            if (q == d) continue;
            while (recorded[q].load(std::memory_order_acquire) < h + 1) {
                if (failed.load()) return 1;
                std::this_thread::yield();
            }
            HIP_TRY(hipStreamWaitEvent(ss[d]->stream, landed[2 * q + (h & 1u)], 0));
        }
        return 0;
    }
};

int run_poismf_multi(const std::vector<int>& devices, real_t* A, real_t* Xr, sparse_ix* Xr_indptr, sparse_ix* Xr_indices, real_t* B, real_t* Xc,
                     sparse_ix* Xc_indptr, sparse_ix* Xc_indices, size_t dimA, size_t dimB, size_t k, const poismf_hip_params& p, size_t numiter)
{
    MultiRun R(devices, dimA, dimB, Xr_indptr, Xc_indptr);
    const size_t nd = R.nd;
    const int method = p.method;
    const bool tn_stop = (method == POISMF_TNCG) && p.early_stop;
    int want_seg = 4;   // segments of the A half (the B shards are a tenth of the size: one)
    if (const char* e = getenv("POISMF_HIP_MULTI_SEGMENTS")) want_seg = std::max(1, atoi(e));

    auto setup = [&](size_t d) -> int {
        if (poismf_hip_session_create(&R.ss[d], devices[d], nullptr, Xr, Xr_indptr, Xr_indices, Xc, Xc_indptr, Xc_indices, dimA, dimB, k,
                                      R.rA[d].lo, R.rA[d].hi, R.rB[d].lo, R.rB[d].hi)) return 1;
        if (poismf_hip_session_set_factors(R.ss[d], A, B)) return 1;
        HIP_TRY(hipSetDevice(devices[d]));
        HIP_TRY(hipStreamCreateWithFlags(&R.copy_stream[d], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&R.seg_done[d], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&R.landed[2 * d], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&R.landed[2 * d + 1], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&R.part_done[d], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&R.part_landed[2 * d], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&R.part_landed[2 * d + 1], hipEventDisableTiming));
        if (want_seg > 1 && poismf_hip_session_set_segments(R.ss[d], 1, want_seg) < 0) return 1;
        for (size_t q = 0; q < nd; q++)   // peer access where the pair allows it (the copies work without, staged by the runtime)
            if (devices[q] != devices[d]) { int can = 0; if (hipDeviceCanAccessPeer(&can, devices[d], devices[q]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devices[q], 0); }
        (void)hipGetLastError();
        return 0;
    };
    // one half on device d: its segments, each followed by the copies of its rows to every peer on the copy stream
    auto half = [&](size_t d, int which, unsigned h, real_t step, real_t cnst_div) -> int {
        poismf_hip_session* s = R.ss[d];
        if (h > 0) {
            if (R.wait_for_peers(d, h - 1)) return 1;
            poismf_hip_session_factors_dirty(s, which ? 0 : 1);   // the fixed factor of this half received rows: its gather copy is re-derived
        }
        // The first stage of this half's column sums is shared (poismf_hip_session_colsum_partial): device d sums its share of the
        // blocks over its whole replica, pushes those partial sums to every peer (copy stream, behind an event of the session stream), and
        // runs the fixed-order second stage once every peer's share has landed -- the unsharded sum bit for bit, 1 / nd of the first stage
        // per device.  Ordering: a peer sends its partials of half h only after it has waited for everybody's rows of half h - 1, i.e.
        // after this device's previous second stage (which read the array) and everything behind it; two events per device alternate,
        // handed over through a counter as the rows' `landed` events are.  Factors below shard_min rows are summed on every device.
        static const size_t shard_min = getenv("POISMF_SHARD_COLSUM_MIN_ROWS") ? (size_t)atoll(getenv("POISMF_SHARD_COLSUM_MIN_ROWS")) : (size_t)262144;
        if (nd > 1 && (which ? dimB : dimA) >= shard_min) {
            const int nb = poismf_hip_session_colsum_blocks(s, which);
            const int b_lo = (int)((size_t)nb * d / nd), b_hi = (int)((size_t)nb * (d + 1) / nd);
            if (poismf_hip_session_colsum_partial(s, which, b_lo, b_hi)) return 1;
            HIP_TRY(hipSetDevice(devices[d]));
            HIP_TRY(hipEventRecord(R.part_done[d], s->stream));
            HIP_TRY(hipStreamWaitEvent(R.copy_stream[d], R.part_done[d], 0));
            const size_t pbytes = (size_t)(b_hi - b_lo) * k * sizeof(real_t);
            for (size_t q = 0; q < nd && pbytes > 0; q++) {
                if (q == d) continue;
                HIP_TRY(hipMemcpyPeerAsync(poismf_hip_session_partials(R.ss[q]) + (size_t)b_lo * k, devices[q],
                                           poismf_hip_session_partials(s) + (size_t)b_lo * k, devices[d], pbytes, R.copy_stream[d]));
            }
            HIP_TRY(hipEventRecord(R.part_landed[2 * d + (h & 1u)], R.copy_stream[d]));
            R.part_recorded[d].store(h + 1, std::memory_order_release);
            for (size_t q = 0; q < nd; q++) {
                if (q == d) continue;
                while (R.part_recorded[q].load(std::memory_order_acquire) < h + 1) {
                    if (R.failed.load()) return 1;
                    std::this_thread::yield();
                }
                HIP_TRY(hipStreamWaitEvent(s->stream, R.part_landed[2 * q + (h & 1u)], 0));
            }
            poismf_hip_session_partials_ready(s);
        }
        const int nseg = (int)s->half[which].segs.size();
        R.unchanged[d] = 0;
        for (int j = 0; j < nseg; j++) {
            if (poismf_hip_half_sweep_segment(s, which, &p, step, cnst_div, j, tn_stop && j == nseg - 1 ? &R.unchanged[d] : nullptr)) return 1;
            HIP_TRY(hipSetDevice(devices[d]));
            size_t lo = 0, hi = 0;
            if (poismf_hip_session_segment_rows(s, which, j, &lo, &hi)) return 1;
            const size_t bytes = (hi - lo) * k * sizeof(real_t);
            HIP_TRY(hipEventRecord(R.seg_done[d], s->stream));
            HIP_TRY(hipStreamWaitEvent(R.copy_stream[d], R.seg_done[d], 0));
            real_t* mine = (which ? s->dA : s->dB) + lo * k;
            for (size_t q = 0; q < nd && bytes > 0; q++) {
                if (q == d) continue;
                real_t* theirs = (which ? R.ss[q]->dA : R.ss[q]->dB) + lo * k;
                HIP_TRY(hipMemcpyPeerAsync(theirs, devices[q], mine, devices[d], bytes, R.copy_stream[d]));
            }
        }
        HIP_TRY(hipEventRecord(R.landed[2 * d + (h & 1u)], R.copy_stream[d]));
        R.recorded[d].store(h + 1, std::memory_order_release);
        return 0;
    };
    auto worker = [&](size_t d) {
        pmf_last_hip_error() = hipSuccess;
        if (setup(d)) R.fail(d);
        R.bar.arrive();
        real_t step_size = p.step_size;
        bool stoppedA = false, stoppedB = false;
        unsigned h = 0;
        for (size_t it = 0; it < numiter; it++) {
            R.bar.arrive([&] { R.stop = g_should_stop != 0 || R.failed.load() != 0; });
            if (R.stop) break;
            const real_t cnst_div = 1. / (1. + 2. * p.l2_reg * step_size);                       // quirk Q6
            for (int which = 0; which < 2; which++) {                                           // B half first (quirk Q5)
                bool& stopped = which ? stoppedA : stoppedB;
                if (!(method == POISMF_TNCG && stopped)) {
                    if (!R.failed.load() && half(d, which, h, step_size, cnst_div)) R.fail(d);
                    h++;
                    if (tn_stop) {                                                              // ref: src/poismf.c:395-403
                        size_t total = 0;
                        R.bar.arrive();
                        for (size_t u : R.unchanged) total += u;
                        R.bar.arrive();   // (everybody has read the counts before the next half resets them)
                        stopped = ((double)total / (double)(which ? dimA : dimB)) >= .95;
                    }
                }
                if (which == 0 && method == POISMF_PG) step_size *= 0.5;                        // ref: :532-533
            }
            if (stoppedA && stoppedB) break;
        }
        // the last half's rows of every peer, then this device is done
        if (!R.failed.load() && h > 0 && R.ss[d] != nullptr) {
            if (R.wait_for_peers(d, h - 1)) R.fail(d);
            else if (hipSetDevice(devices[d]) != hipSuccess || hipStreamSynchronize(R.ss[d]->stream) != hipSuccess ||
                     hipStreamSynchronize(R.copy_stream[d]) != hipSuccess) { pmf_last_hip_error() = hipGetLastError(); R.fail(d); }
            else if (team_check(R.ss[d])) R.fail(d);
        }
        R.bar.arrive();
    };
    {
        std::vector<std::thread> th;
        for (size_t d = 1; d < nd; d++) th.emplace_back(worker, d);
        worker(0);
        for (auto& t : th) t.join();
    }
    int rc = R.failed.load() ? 1 : 0;
    if (rc) {   // the failing worker's error is what the caller reports (the workers' thread-local slots are gone)
        pmf_last_hip_error() = hipSuccess;
        for (hipError_t e : R.err) if (e != hipSuccess) { pmf_last_hip_error() = e; break; }
    }
    if (!rc) rc = poismf_hip_session_get_factors(R.ss[0], A, B);   // every replica holds the same bits
    for (size_t d = 0; d < nd; d++) {
        (void)hipSetDevice(devices[d]);
        if (R.copy_stream[d]) { (void)hipStreamSynchronize(R.copy_stream[d]); (void)hipStreamDestroy(R.copy_stream[d]); }
        if (R.seg_done[d]) (void)hipEventDestroy(R.seg_done[d]);
        for (int e = 0; e < 2; e++) if (R.landed[2 * d + e]) (void)hipEventDestroy(R.landed[2 * d + e]);
        if (R.part_done[d]) (void)hipEventDestroy(R.part_done[d]);
        for (int e = 0; e < 2; e++) if (R.part_landed[2 * d + e]) (void)hipEventDestroy(R.part_landed[2 * d + e]);
        poismf_hip_session_destroy(R.ss[d]);
    }
    return rc ? 1 : 0;
}

// POISMF_HIP_DEVICES: comma-separated device ids; empty / one entry: the single-device path
std::vector<int> devices_from_env()
{
    std::vector<int> out;
    const char* e = getenv("POISMF_HIP_DEVICES");
    if (e == nullptr) return out;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return out;
    for (const char* q = e; *q;) {
        char* end = nullptr;
        const long v = strtol(q, &end, 10);
        if (end == q) break;
        if (v >= 0 && v < n) out.push_back((int)v);
        q = (*end == ',') ? end + 1 : end;
        if (*end != ',' && *end != 0) break;
    }
    return out;
}

}  // namespace
}  // extern "C++"

// run_poismf's loop on a session that already holds X and the starting factors (PoisMF.fit keeps the CSR / CSC it
// built on the device and never takes them through host memory).  Same return codes as run_poismf.
int poismf_hip_session_run(poismf_hip_session* s, const poismf_hip_params* p, size_t numiter, int handle_interrupt)
{
    SigintScope sig;
    sig.enter();
    pmf_last_hip_error() = hipSuccess;
    int ret_code = 0;
    if (hipSetDevice(s->device) != hipSuccess || run_alternation(s, *p, numiter)) {
        pmf_report_failure();
        ret_code = 1;
    }
    return sig.leave(ret_code, handle_interrupt != 0);
}

int run_poismf(real_t* A, real_t* Xr, sparse_ix* Xr_indptr, sparse_ix* Xr_indices, real_t* B, real_t* Xc,
               sparse_ix* Xc_indptr, sparse_ix* Xc_indices, const size_t dimA, const size_t dimB, const size_t k,
               const real_t l2_reg, const real_t l1_reg, const real_t w_mult, real_t step_size, const int method,
               const bool limit_step, const size_t numiter, const size_t maxupd, const bool early_stop,
               const bool reuse_prev, const bool handle_interrupt, const int nthreads)
{
    (void)nthreads;
    SigintScope sig;
    sig.enter();
    pmf_last_hip_error() = hipSuccess;

    int ret_code = 0;
    poismf_hip_session* s = nullptr;
    int device = 0;
    if (const char* e = getenv("POISMF_HIP_DEVICE")) device = atoi(e);

    poismf_hip_params p;
    p.l2_reg = l2_reg; p.l1_reg = l1_reg; p.w_mult = w_mult; p.step_size = step_size;
    p.method = method; p.limit_step = limit_step; p.maxupd = maxupd;
    p.early_stop = early_stop; p.reuse_prev = reuse_prev;
    {   // several GPUs of this node (POISMF_HIP_DEVICES=0,1,..): same call, same results, rows sharded over the devices
        const std::vector<int> devs = devices_from_env();
        if (devs.size() > 1) {
            if (run_poismf_multi(devs, A, Xr, Xr_indptr, Xr_indices, B, Xc, Xc_indptr, Xc_indices, dimA, dimB, k, p, numiter)) {
                pmf_report_failure();
                ret_code = 1;
            }
            return sig.leave(ret_code, handle_interrupt);
        }
        if (devs.size() == 1) device = devs[0];
    }
    // POISMF_HIP_VERBOSE=1: wall time of the phases of this call on stderr (development aid, scripts/time_abi.py)
    static const bool verbose = getenv("POISMF_HIP_VERBOSE") != nullptr;
    double t[5] = { 0, 0, 0, 0, 0 };
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6; };
    t[0] = now();
    // The session is put together in the order the first iteration needs things: the B side's matrix (CSC) and the factors first; the A
    // side's matrix (CSR: a third of the call's PCIe bytes) is uploaded -- pinned chunks, second stream -- and its rows are sorted while
    // the first B half runs (round 4).  poismf_hip_session_create does the same two build_half calls back to back.
    pmf_tl(nullptr);
    s = session_alloc(device, nullptr, dimA, dimB, k);
    pmf_tl("session allocated (factors, streams)");
    bool bad = s == nullptr;
    HalfPending pendB;   // (the B side's row sort runs on the device while the host threads stage the factors)
    bad = bad || build_half(s->half[0], s->stream, Xc, Xc_indptr, Xc_indices, dimB, dimA, 0, dimB, device, &pendB);
    t[1] = now();
    bad = bad || poismf_hip_session_set_factors(s, A, B);
    pmf_tl("factors handed to the DMA queue");
    if (!bad) bad = finish_half_collect(s->half[0], s->stream, pendB) != 0;
    else if (s != nullptr) { pmf_free(pendB.d_flag, s->stream); pmf_free(pendB.d_len, s->stream); }
    pmf_tl("B side: lengths back, bins cut");
    t[2] = now();
    double t_csr = 0;
    const std::function<int()> upload_csr = [&]() -> int {
        const double t0 = now();
        HIP_TRY(hipSetDevice(device));
        if (build_half(s->half[1], s->aux_stream, Xr, Xr_indptr, Xr_indices, dimA, dimB, 0, dimA, device)) return 1;
        HIP_TRY(hipStreamSynchronize(s->aux_stream));
        t_csr = now() - t0;
        return 0;
    };
    static const bool no_overlap = getenv("POISMF_HIP_NO_UPLOAD_OVERLAP") != nullptr;   // testing knob: the whole matrix before anything runs
    if (!bad && (no_overlap || numiter == 0)) bad = upload_csr() != 0;
    bad = bad || run_alternation(s, p, numiter, (no_overlap || numiter == 0) ? nullptr : &upload_csr);
    t[3] = now();
    pmf_tl("iterations done");
    bad = bad || poismf_hip_session_get_factors(s, A, B);
    pmf_tl("factors back");
    t[4] = now();
    if (bad) {
        pmf_report_failure();   // "Error: out of memory." (ref: :501) only when it was one
        ret_code = 1;
    }
    poismf_hip_session_destroy(s);
    pmf_tl("session destroyed");
    if (verbose)
        fprintf(stderr, "run_poismf: session + B side of X (upload, sort rows) %.2f ms, factors up %.2f ms, %zu iterations %.2f ms (of which the A "
                        "side of X, uploaded under the first B half: %.2f ms), factors down %.2f ms, teardown %.2f ms\n", t[1] - t[0], t[2] - t[1],
                numiter, t[3] - t[2], t_csr, t[4] - t[3], now() - t[4]);
    return sig.leave(ret_code, handle_interrupt);
}

// -------------------------------------------------------------------------------------------------
// factors_multiple: latent factors of new rows with B fixed          ref: src/pred.c:66-199
// -------------------------------------------------------------------------------------------------
static int factors_multiple_impl(real_t* A, real_t* B, real_t* Bsum, real_t* Amean, real_t* Xr, sparse_ix* Xr_indptr,
                                 sparse_ix* Xr_indices, int k, size_t dimA, real_t l2_reg, real_t w_mult, real_t step_size,
                                 size_t niter, size_t maxupd, int method, bool limit_step, bool reuse_mean, unsigned* decisions);
int factors_multiple(real_t* A, real_t* B, real_t* Bsum, real_t* Amean, real_t* Xr, sparse_ix* Xr_indptr,
                     sparse_ix* Xr_indices, int k, size_t dimA, real_t l2_reg, real_t w_mult, real_t step_size,
                     size_t niter, size_t maxupd, int method, bool limit_step, bool reuse_mean, int nthreads)
{
    (void)nthreads;
    return factors_multiple_impl(A, B, Bsum, Amean, Xr, Xr_indptr, Xr_indices, k, dimA, l2_reg, w_mult, step_size, niter, maxupd, method,
                                 limit_step, reuse_mean, nullptr);
}
// Testing aid (G1): the device's own objective and gradient wrappers at a given point, row by row, through whatever engine a CG
// half-sweep would use for rows of that length (plan.hpp, K_EVAL).  which = 0: fun_single + grad_single (ref: src/poismf.c:194-240);
// 1: fun_and_grad (ref: :242-273).  G [dimA x k] gets the gradients, f [dimA] the function values; every row is evaluated at `point`.
int poismf_hip_debug_row_eval(real_t* G, double* f, real_t* B, real_t* Bsum, real_t* point, real_t* Xr, sparse_ix* Xr_indptr,
                              sparse_ix* Xr_indices, int k, size_t dimA, real_t l2_reg, real_t w_mult, int which)
{
    std::vector<unsigned> dec;
    try { dec.assign(2 * dimA, 0u); } catch (const std::bad_alloc&) { return 1; }
    const int rc = factors_multiple_impl(G, B, Bsum, point, Xr, Xr_indptr, Xr_indices, k, dimA, l2_reg, w_mult, (real_t)1e-7, 1, which ? 1 : 0,
                                         POISMF_EVAL, true, true, dec.data());
    if (rc) return rc;
    for (size_t r = 0; r < dimA; r++) {
        const unsigned long long b = ((unsigned long long)dec[2 * r + 1] << 32) | dec[2 * r];
        memcpy(&f[r], &b, sizeof(double));
    }
    return 0;
}
// Testing aid: factors_multiple that also hands back every row's solver decisions (2 words per row, see
// poismf_hip_session_decisions) -- how the golden single-row fixtures pin the device's iteration / evaluation counts.
int poismf_hip_factors_multiple_decisions(real_t* A, real_t* B, real_t* Bsum, real_t* Amean, real_t* Xr, sparse_ix* Xr_indptr,
                                          sparse_ix* Xr_indices, int k, size_t dimA, real_t l2_reg, real_t w_mult, real_t step_size,
                                          size_t niter, size_t maxupd, int method, bool limit_step, bool reuse_mean, unsigned* decisions)
{
    return factors_multiple_impl(A, B, Bsum, Amean, Xr, Xr_indptr, Xr_indices, k, dimA, l2_reg, w_mult, step_size, niter, maxupd, method,
                                 limit_step, reuse_mean, decisions);
}
static int factors_multiple_impl(real_t* A, real_t* B, real_t* Bsum, real_t* Amean, real_t* Xr, sparse_ix* Xr_indptr,
                                 sparse_ix* Xr_indices, int k, size_t dimA, real_t l2_reg, real_t w_mult, real_t step_size,
                                 size_t niter, size_t maxupd, int method, bool limit_step, bool reuse_mean, unsigned* decisions)
{
    pmf_last_hip_error() = hipSuccess;   // (what this call reports on failure is this call's error, not an earlier call's)
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
        if (decisions != nullptr) poismf_hip_session_profile(s, 1);
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
                rc = half_sweep_impl(s, 1, &p, step_size, cnst_div, nullptr, bs.data(), -step0, weighted ? -step_size : (real_t)1);
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
        if (!rc) rc = team_check(s);
        if (!rc && decisions != nullptr) rc = poismf_hip_session_decisions(s, 1, decisions, dimA);
    }
    poismf_hip_session_destroy(s);
    if (rc) pmf_report_failure();
    return rc ? 1 : 0;
}

}  // extern "C"
