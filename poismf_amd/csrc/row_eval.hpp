// row_eval.hpp -- the per-row evaluation engine shared by the three solvers.
//
// One wavefront owns one output row r of the factor being updated (M) against the fixed opposing
// factor F.  The rows F[ind_j] named by the row's nonzeros are gathered ONCE from HBM/L2 into an LDS
// tile with 16-byte global loads (4-/8-byte aligned; k = 50 fp32: 13 slots per 200-byte row, so every
// load instruction moves 64 slots = 1 KiB) and every inner pass of the solver then runs from LDS.
//
// Register layout of every k-vector ("slot layout"): a lane holds whole 16-byte SLOTS (4 floats / 2
// doubles), lane g of a group of G lanes holds slot g (+ G, + 2G .. when the row has more than 64
// slots).  G = 16, 32 or 64 is the smallest power of two that covers the row, and the wave holds
// JG = 64 / G identical COPIES of every vector, one per group.  Consequences:
//   * k-length reductions are 4 DPP steps inside a 16-lane row (+1 / +3 cross-row combines for G = 32 / 64);
//   * phase 2 below reads the tile with ds_read_b128 and lets the JG groups work on JG nonzeros at once.
//
//   phase 1  lane <-> nonzero : pred_j = T[j,:] . a   (ds_read_b128 down the lane's own tile row; the row
//            stride is an ODD number of slots, so the 16 lanes of a b128 service group hit 16 distinct slots;
//            `a` is broadcast from LDS), then coef_j = +-x_j / pred_j  and / or  x_j log(pred_j)
//   phase 2  lane <-> (nonzero group, slot) : acc[slot] += coef_j * T[j, slot]  for the group's nonzeros
//            j = jg, jg + JG, ..; the JG partial sums are combined once per evaluation
//
// which is what the reference does per nonzero with one ddot + one daxpy
// (ref: src/poismf.c:126-133 calc_grad_pgd, :194-208 calc_fun_single, :210-240 calc_grad_single[_w],
// :242-273 calc_fun_and_grad), with no per-nonzero cross-lane reduction and no re-gather.
// Rows that do not fit the tile (cap) are streamed chunk by chunk on every pass instead.
#pragma once
#include "wave_ops.hpp"

// -DPMF_PROBE (development): shader-clock stamps of the phases of ONE evaluation (the 4th of a row) in one workgroup, written
// over the per-row evaluation counters; scripts/probes/probe_phases.py reads them back.
#ifdef PMF_PROBE
#define PMF_STAMP(ev, i)                                                                                     \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if ((ev).probe != nullptr && (ev).n_eval == 4 && (ev).lane == 0) (ev).probe[i] = (unsigned)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#else
#define PMF_STAMP(ev, i) do {} while (0)
#endif

namespace pmf {

// ---- teams: one row over the register tiles of several CUs (reg_eval.hpp, M_ > 1) ------------------------------------------
// For rows whose tile does not fit one CU's registers but does fit a few (k = 50 fp64, ~1000 nonzeros: 400 KB), where the
// alternative is to stream the tile from L2 / HBM once per evaluation.  Workgroups of TEAM_NW waves, 28 or 32 steps each; the
// partial gradients / log-likelihood sums of an evaluation cross CUs through tagged 8-byte granules in global memory.
constexpr int TEAM_NW = 4, TEAM_M_MAX = 4;   // one wave per SIMD: CG's state next to a two-slot tile takes the 512 registers
// layout of HalfArgs::team_buf, in 8-byte words: [0, 8) arrival counters per XCD, [8] error word, then TEAM_SLOTS teams of
// TEAM_WORDS words: { mailbox[2], here[TEAM_M_MAX], exchange[2 parities][TEAM_M_MAX members][TEAM_GRAN granules] }
constexpr unsigned TEAM_SLOTS_PER_XCD = 128, TEAM_SLOTS = 8 * TEAM_SLOTS_PER_XCD;
constexpr int TEAM_SC = 4;                       // scalars per exchange (a line search asks for that many trial steps at once)
constexpr unsigned TEAM_GRAN = 2 * 64 + 2 * TEAM_SC;   // one k-vector of <= 64 doubles and the scalars, as (low, high) halves
constexpr unsigned TEAM_HEAD_WORDS = 16;
constexpr unsigned TEAM_WORDS = 2 + TEAM_M_MAX + 2 * TEAM_M_MAX * TEAM_GRAN + 2;   // (+2: keeps teams 16-byte aligned)
constexpr unsigned long long TEAM_BUF_BYTES = 8ull * (TEAM_HEAD_WORDS + (unsigned long long)TEAM_SLOTS * TEAM_WORDS);
constexpr unsigned TEAM_SPIN_LIMIT = 1u << 20;   // polls (~1 us each) before a member gives the launch up

// ---- giant rows: one STREAMED row over the LDS tiles of many CUs (RowEval, TM = true; round 5) -------------------------------
// The power-law tail -- config C5's 60 item rows of 8 k .. 145 k nonzeros -- never fits on chip: it is re-gathered for every one of
// TNC's evaluations, and ONE eight-wave workgroup per row moves ~70 GB/s (the biggest row alone took 120 ms of the half-sweep).
// A team of GT_M workgroups (one per CU, formed in arrival order) takes one row: member m streams nonzeros [m S, (m + 1) S), S =
// ceil(nnz / GT_M) rounded up to whole 64s, through its eight waves exactly as a single workgroup streams a whole row; per
// evaluation the members' partial gradients and log-likelihood sums cross CUs as tagged 8-byte granules (as the register teams'
// do) and are added in member order, so every wave of every member keeps the same bits and takes the same branches.
// Buffer (8-byte words): a header line ([0]: arrival counter), then per team { a line with mailbox[2], exchange[2 parities][members][GT_GRAN words] };
// a member's GT_GRAN words are whole 128-byte lines: its values' granules, the scalar's, and pad granules (team_sum).
constexpr int GT_M = 32;                           // members of a giant-row team (a function of nothing: a row's arithmetic must not depend on its launch)
constexpr int GT_VALS = 258;                       // doubles per exchange: a k-vector of up to 256 elements (k <= 256 fp64 / 512 fp32: two exchanges' worth is never needed) + the sum
constexpr unsigned GT_PAD_AT = 2 * GT_VALS;          // first of a member's pad granules: where lanes that carry no value of their own publish (team_sum)
constexpr unsigned GT_PAD_WORDS = 32;
constexpr unsigned GT_GRAN = (2 * GT_VALS + GT_PAD_WORDS + 15) / 16 * 16;   // words per member and set, whole 128-byte lines (no line is written by two members)
constexpr unsigned GT_HEAD_WORDS = 16;
constexpr unsigned GT_MAIL_WORDS = 16;   // a team's mailbox[2], a line of its own
constexpr unsigned GT_TEAM_WORDS = GT_MAIL_WORDS + 2 * GT_M * GT_GRAN;
__host__ __device__ constexpr unsigned gt_team_words(unsigned members) { return GT_MAIL_WORDS + 2u * members * GT_GRAN; }   // (teams of other sizes share the buffer: lane_eval.hpp, TM_)
constexpr unsigned GT_TEAMS_MAX = 16;
constexpr unsigned long long GT_BUF_BYTES = 8ull * (GT_HEAD_WORDS + (unsigned long long)GT_TEAMS_MAX * GT_TEAM_WORDS);
__device__ __forceinline__ void gt_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long gt_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One exchange of a team of `M` workgroups, carried out by ONE wave of each member (all 64 lanes call it): lane-local values val[i] (valid[i]:
// this lane carries one; idx[i] < GT_VALS - 1: its place in the member's vector) and one wave-uniform scalar are published as tagged granules,
// the other members' are collected and everything is added in MEMBER ORDER (every member ends with the same bits).  Two alternating sets of
// granules per member (a member can be at most one exchange ahead of the slowest).  The collection is pipelined four members deep: the loads of
// members m + 1 .. m + 3 are in flight while member m's tags are checked (a team of 32 took ~25 us per exchange one member at a time); a member
// whose granules are not there yet is polled on its own.  `words`: the team's area { mailbox[2], granules[2 parities][M][GT_GRAN] }.
// Returns false when an exchange timed out (the launch's error word is set: everybody leaves, the host re-runs the launch's rows).
template <int NV>
__device__ __forceinline__ bool team_sum(unsigned long long* words, int M, int member, unsigned seq, unsigned* err, unsigned spin_limit, int lane,
                                         const int (&idx)[NV], const bool (&valid)[NV], double (&val)[NV], double& scal)
{
    static_assert(2 * NV <= (int)GT_PAD_WORDS, "one pad granule per value");
    const unsigned long long tag = (unsigned long long)seq << 32;
    unsigned long long* slots = words + GT_MAIL_WORDS + (size_t)(seq & 1u) * (size_t)M * GT_GRAN;
    unsigned long long* mine = slots + (size_t)member * GT_GRAN;
    // NO lane-dependent branch in here: a lane without a value of its own publishes and collects a PAD granule (tagged like the others, so
    // its polls succeed with everybody else's) and the scalar is stored and loaded by all 64 lanes.  Under a 512-register tile the compiler
    // spills around this code, and a spill placed inside a lane-divergent region saves only the active lanes' copies: the others came back
    // with the previous exchange's values (measured: a member re-publishing its previous sum -- rarely, box to box).
    unsigned at[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) at[i] = valid[i] ? 2u * (unsigned)idx[i] : GT_PAD_AT + 2u * (unsigned)i;
    constexpr unsigned SCAL_AT = 2u * (GT_VALS - 1);
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, val[i]);
        gt_store(mine + at[i], (b & 0xffffffffull) | tag);
        gt_store(mine + at[i] + 1, (b >> 32) | tag);
    }
    {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, scal);
        gt_store(mine + SCAL_AT, (b & 0xffffffffull) | tag);
        gt_store(mine + SCAL_AT + 1, (b >> 32) | tag);
    }
    constexpr int DEPTH = 4;
    unsigned long long lo[DEPTH][NV], hi[DEPTH][NV], l0[DEPTH], l1[DEPTH];
    auto request = [&](int m, int slot) {   // (wave-uniform condition; own member and past the last: nothing to load)
        if (m != member && m < M) {
            const unsigned long long* theirs = slots + (size_t)m * GT_GRAN;
#pragma unroll
            for (int i = 0; i < NV; i++) { lo[slot][i] = gt_load(theirs + at[i]); hi[slot][i] = gt_load(theirs + at[i] + 1); }
            l0[slot] = gt_load(theirs + SCAL_AT); l1[slot] = gt_load(theirs + SCAL_AT + 1);
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++) { lo[slot][i] = tag; hi[slot][i] = tag; }
            l0[slot] = tag; l1[slot] = tag;
        }
    };
    double sum[NV], lt = 0.0;
#pragma unroll
    for (int i = 0; i < NV; i++) sum[i] = 0.0;
    bool dead = false;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) request(d, d);
    for (int m0 = 0; m0 < M; m0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const int m = m0 + d;
            request(m + DEPTH - 1, (d + DEPTH - 1) % DEPTH);
            if (m < M) {
                double pv[NV], pl = scal;
#pragma unroll
                for (int i = 0; i < NV; i++) pv[i] = val[i];
                if (m != member && !dead) {
                    unsigned spins = 0;
                    for (;;) {
                        bool ok = (l0[d] >> 32) == seq && (l1[d] >> 32) == seq;
#pragma unroll
                        for (int i = 0; i < NV; i++) ok = ok && (lo[d][i] >> 32) == seq && (hi[d][i] >> 32) == seq;
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 255u) == 0 && (spins > spin_limit || uniform(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0)) {
                            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            dead = true;
                            break;
                        }
                        request(m, d);
                    }
#pragma unroll
                    for (int i = 0; i < NV; i++) pv[i] = __builtin_bit_cast(double, (lo[d][i] & 0xffffffffull) | (hi[d][i] << 32));
                    pl = uniform(__builtin_bit_cast(double, (l0[d] & 0xffffffffull) | (l1[d] << 32)));
                }
#pragma unroll
                for (int i = 0; i < NV; i++) sum[i] = m == 0 ? pv[i] : sum[i] + pv[i];
                lt = m == 0 ? pl : lt + pl;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; i++) val[i] = sum[i];
    scal = lt;
    return !dead;
}


// The same exchange OUT OF LINE, for kernels whose registers are full (the lane engine's resident teams, 512 registers + scratch): compiled on
// its own it has no spills, and what the caller spills around the call it spills at the call site, where every lane of the wave is active.
template <int NV> struct TeamVals { double v[NV]; double s; int idx[NV]; bool valid[NV]; bool ok; };
template <int NV>
__device__ __attribute__((noinline)) TeamVals<NV> team_sum_call(unsigned long long* words, int M, int member, unsigned seq, unsigned* err, unsigned spin_limit, int lane,
                                                                TeamVals<NV> x)
{
    x.ok = team_sum<NV>(words, M, member, seq, err, spin_limit, lane, x.idx, x.valid, x.v, x.s);
    return x;
}

// Optional phase timers (build with -DPMF_TIMING): per-wave shader-clock totals of the phases of row_eval,
// added to a global array at kernel exit.  Slots: 0 gather, 1 phase 1, 2 coef/div, 3 phase 2, 4 combine, 5 whole kernel.
#ifdef PMF_TIMING
static __device__ unsigned long long g_pmf_timing[8];   // one per translation unit: the row kernels' unit reads its own
#define PMF_T0(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define PMF_T1(slot, v) tacc[slot] += __builtin_amdgcn_s_memtime() - v
#else
#define PMF_T0(v)
#define PMF_T1(slot, v)
#endif

// unroll depths of the two LDS phases (slots per block in phase 1, steps per block in phase 2)
#ifndef PMF_P1_BLOCK
#define PMF_P1_BLOCK 8
#endif
#ifndef PMF_P2_BLOCK
#define PMF_P2_BLOCK 16
#endif
#ifndef PMF_PRE
#define PMF_PRE 19
#endif

template <class T> struct Slot;
template <> struct Slot<float> {
    static constexpr int N = 4;
    struct __attribute__((packed, aligned(4))) U { float v[4]; };   // 16 B at 4-byte alignment
    struct __attribute__((aligned(16))) A { float v[4]; };
};
template <> struct Slot<double> {
    static constexpr int N = 2;
    struct __attribute__((packed, aligned(8))) U { double v[2]; };  // 16 B at 8-byte alignment
    struct __attribute__((aligned(16))) A { double v[2]; };
};

// Geometry of one launch (host fills it in; see plan_geom in poismf_hip.hip).
struct TileGeom {
    int k;         // factor dimension
    int s_load;    // 16-byte slots actually holding data per factor row = ceil(k*sizeof(T)/16)
    int s_stride;  // LDS row stride in slots = s_load | 1 (odd: conflict-free ds_read_b128 down a column)
    int cap;       // nonzeros the tile can hold
    int resident;  // 1: every row of this launch has nnz <= cap, gather once per row
    int group;     // G: lanes per vector copy (16, 32 or 64)
    int pq_cap;    // nonzeros for which the two per-nonzero prediction caches (T.x, T.d) fit in LDS; 0 = no cache
    unsigned zero_row;  // index of the all-zero row the session keeps behind the factor F (= its row count)
    int ldF;       // elements between consecutive rows of the gathered factor (k, or more in the line-padded copy)
    int prefetch;  // 1: streamed rows keep a second set of index / value buffers (next chunk's tile is requested early)
};

// Problem constants of one half-sweep that the solvers need.
template <class T> struct RowParams {
    T l2, w;
    T step, cnst_div, neg_step;  // PG (step already multiplied by w, ref: src/poismf.c:151)
    T neg_step2;                 // PG, w != 1: a second scale applied after neg_step (factors_multiple), 1 = none
    int maxupd;
    int limit_step;
    int max_cg_it;               // TNC: max(1, min(50, k/2)), ref: src/poismf.c:342
    int x_pos;                   // 1: every stored value of this half's matrix is > 0 (counts): the data term of the objective is
                                 // concave along a line, which lets CG's line search skip trial steps that are certain to fail
};

__host__ __device__ inline size_t lds_bytes_per_wave(const TileGeom& g, size_t sizeof_real)
{
    size_t b = (size_t)g.cap * g.s_stride * 16;          // tile
    b += (size_t)g.s_load * 16;                          // current point a (padded with zeros)
    b += (((size_t)g.cap * sizeof_real) + 15) / 16 * 16; // x_j
    b += (((size_t)g.cap * 4) + 15) / 16 * 16;           // ind_j
    b += 64 * sizeof_real;                               // coef_j of the 64 nonzeros in flight
    b += (((size_t)2 * g.pq_cap * sizeof_real) + 15) / 16 * 16;  // cached predictions p_j = T_j.x and q_j = T_j.d
    if (g.prefetch)                                      // x_j, ind_j of the NEXT chunk
        b += (((size_t)g.cap * sizeof_real) + 15) / 16 * 16 + (((size_t)g.cap * 4) + 15) / 16 * 16;
    return b;
}

__host__ __device__ inline size_t lds_bytes_per_block(const TileGeom& g, size_t sizeof_real, int nw)
{
    size_t b = (size_t)nw * lds_bytes_per_wave(g, sizeof_real);
    // two sets of cross-wave scratch (used alternately: one barrier per evaluation) + the row-queue broadcast word
    if (nw > 1) b += 2 * (16 * (((size_t)nw * sizeof(double) + 15) / 16) + (size_t)nw * g.s_load * 16) + 16;
    return b;
}

// NC = elements per lane = slots per lane (NS) x elements per slot.
// SL > 0 fixes the number of 16-byte slots per factor row at compile time (specialisations for the k values of
// the BASELINE configs): the row stride, group size and every LDS offset of the two phases then fold into
// instruction immediates, which removes most of the address arithmetic (measured: 85 % of the issued
// instructions of the generic phase-2 loop were address / mask bookkeeping).  SL = 0 is the generic kernel.
// NW > 1: NW wavefronts (one workgroup) cooperate on ONE row -- the long-row path for the power-law tail.  Every
// wave keeps its own full copy of the solver state and runs the same wave-uniform control flow; inside an
// evaluation wave w streams chunks w, w + NW, ... of the row through its private LDS tile, and the NW partial
// results are combined through LDS in a fixed order behind a workgroup barrier, so all copies stay bit-identical.
// What it buys is memory-level parallelism: one wave keeps ~8 KiB of gathers in flight (~4 GB/s at ~2 us of
// latency), eight waves on a CU approach that CU's ~24 GB/s.
// PF: streamed rows request the NEXT chunk's factor rows (into registers) before they work on the current chunk and drop
// them into the tile afterwards, so that a chunk's gather overlaps the previous chunk's two phases.
template <class T, int NC, int SL = 0, int NW = 1, bool PF = false, bool TM = false> struct RowEval {
    using SA = typename Slot<T>::A;
    using SU = typename Slot<T>::U;
    static constexpr int SN = Slot<T>::N;
    static constexpr int NS = NC / SN;
    static constexpr bool PIPELINED = false;  // sweep_rows: no cross-row prefetch (the LDS tile has one set of index buffers)
    static constexpr int PIPE_MW = 1;             // (lane_eval.hpp: whether multi-wave rows take the pipeline depends on the solver)
    static constexpr bool FUSED_SUMS = false;     // (lane_eval.hpp reduces the solvers' groups of dot products together)
    static constexpr bool PREFETCH = false;       // (lane_eval.hpp can request the next row's tile while this one is solved)
    static constexpr bool MAY_CACHE = true;   // cached CG line search where the launch geometry has room for it (pq_cap)
    static constexpr int PRE = PMF_PRE;       // 16-byte slots per lane a prefetched chunk may take (19: 1216 slots = 19 KiB)
    static_assert(NC % SN == 0, "a lane holds whole 16-byte slots");

    // LDS carve-out of this wave
    SA* tile;
    SA* avec;
    T* xb;
    unsigned* idxb;
    T* coefb;
    T* pbuf;  // p_j = F[ind_j,:] . x   for every nonzero of the row (CG line-search cache)
    T* qbuf;  // q_j = F[ind_j,:] . d
    int pq_cap;
    // streamed rows: indices / values of the NEXT chunk travel in registers while the current chunk is processed
    unsigned m_idx[2];
    T m_x[2];
    unsigned meta_c0; // first nonzero of the chunk whose indices / values sit in idxb / xb (0xffffffff: none)
    unsigned tile_c0; // ... of the chunk whose factor rows sit in the tile          (PF only)
    unsigned m_c0;    // ... of the chunk whose indices / values sit in m_idx / m_x  (PF only)
    T* xb2;           // PF: values / indices of the chunk whose tile has been requested
    unsigned* idxb2;
    bool pf_on;
    unsigned char* red_base;  // NW > 1: two sets of { [NW] partial log-likelihood sums, [NW][s_load] slots of partial gradients }
    int red_sel, red_bytes;   // the set the next combine_waves uses; bytes per set
    int wid;
    int member = 0;                    // TM: this workgroup's place in its giant-row team (0 otherwise)
    int tm_M = 1;                      // TM: members of the team
    unsigned tm_seq = 0;               // TM: exchanges so far
    unsigned long long* tm_words = nullptr;   // TM: this team's exchange area
    unsigned* tm_err = nullptr;        // TM: != 0: some exchange of this launch timed out, give up
    unsigned tm_spin = TEAM_SPIN_LIMIT;
    static_assert(!TM || NW > 1, "giant-row teams are teams of multi-wave workgroups");
    static constexpr unsigned TEAM_ROUND = 64;   // a member's share of a row is a whole number of these
    // launch constants
    const T* F;
    int k, ldF, s_load, s_stride, cap, tail;
    bool resident;
    int lane, G, JG, g, jg;
    int gj0, gt0, gdj, gdt;  // lane -> (nonzero, slot) walk of the gather, advanced 64 slots at a time
    int elem[NC];            // factor dimension held in element i of this lane
    bool act[NC];            // elem[i] < k
    int slotq[NS];           // slot index held in slot s of this lane, clamped into the row
    bool slot_on[NS];
    // current row
    const unsigned* ind;
    const T* val;
    unsigned nnz;
    unsigned n_eval;  // passes over the row's tile since the caller last reset it (wave-uniform; reporting only)
#ifdef PMF_PROBE
    unsigned* probe = nullptr;
#endif
#ifdef PMF_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    __device__ __forceinline__ void init(const TileGeom& geo, const T* F_, unsigned char* smem)
    {
        lane = lane_id();
        F = F_;
        k = geo.k; ldF = geo.ldF; cap = geo.cap; resident = geo.resident != 0;
        if constexpr (SL > 0) {  // compile-time geometry (the host only launches this instance when it matches)
            s_load = SL; s_stride = SL | 1; G = SL <= 16 ? 16 : (SL <= 32 ? 32 : 64);
        } else {
            s_load = geo.s_load; s_stride = geo.s_stride; G = geo.group;
        }
        JG = WAVE / G;
        g = lane & (G - 1); jg = lane / G;
        tail = k - (s_load - 1) * SN;  // valid elements in the last slot of a factor row (1..SN)
        wid = NW > 1 ? (int)(threadIdx.x / WAVE) : 0;
        const size_t wave_bytes = lds_bytes_per_wave(geo, sizeof(T));
        unsigned char* p = smem + (size_t)wid * wave_bytes;
        red_base = smem + (size_t)NW * wave_bytes;
        red_sel = 0;
        red_bytes = (int)(16 * ((NW * sizeof(double) + 15) / 16)) + NW * s_load * 16;
        tile = (SA*)p; p += (size_t)cap * s_stride * 16;
        avec = (SA*)p; p += (size_t)s_load * 16;
        xb = (T*)p; p += (((size_t)cap * sizeof(T)) + 15) / 16 * 16;
        idxb = (unsigned*)p; p += (((size_t)cap * 4) + 15) / 16 * 16;
        coefb = (T*)p; p += 64 * sizeof(T);
        pq_cap = geo.pq_cap;
        pbuf = (T*)p; qbuf = pbuf + pq_cap;
        p += (((size_t)2 * pq_cap * sizeof(T)) + 15) / 16 * 16;
        pf_on = PF && geo.prefetch != 0 && !resident && cap <= 2 * WAVE && cap * s_load <= PRE * WAVE;
        xb2 = (T*)p; p += (((size_t)cap * sizeof(T)) + 15) / 16 * 16;
        idxb2 = (unsigned*)p;
        ticket_word = (unsigned*)(smem + lds_bytes_per_block(geo, sizeof(T), NW) - 16);
        gj0 = lane / s_load; gt0 = lane % s_load;
        gdj = WAVE / s_load; gdt = WAVE % s_load;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int q = g + G * s;
            slot_on[s] = q < s_load;
            slotq[s] = slot_on[s] ? q : 0;
#pragma unroll
            for (int e = 0; e < SN; e++) {
                elem[s * SN + e] = q * SN + e;
                act[s * SN + e] = q * SN + e < k;
            }
        }
    }

    // ---- k-length vector helpers on the slot layout ------------------------------------------------
    // reduce over ONE copy of a vector; every lane receives the (wave-uniform) result
    template <class Op, class V> __device__ __forceinline__ V reduce(V x) const
    {
        x = Op::f(x, dpp_mov<0xB1>(x));
        x = Op::f(x, dpp_mov<0x4E>(x));
        x = Op::f(x, dpp_mov<0x141>(x));
        x = Op::f(x, dpp_mov<0x140>(x));
        if (G == 16) return uniform(x);
        if (G == 32) return uniform(Op::f(read_lane(x, 0), read_lane(x, 16)));
        const V r0 = read_lane(x, 0), r1 = read_lane(x, 16), r2 = read_lane(x, 32), r3 = read_lane(x, 48);
        return uniform(Op::f(Op::f(r0, r1), Op::f(r2, r3)));
    }
    template <class V> __device__ __forceinline__ V rsum(V x) const { return reduce<OpSum>(x); }
    template <class V> __device__ __forceinline__ V rmin(V x) const { return reduce<OpMin>(x); }
    template <class V> __device__ __forceinline__ V rmax(V x) const { return reduce<OpMax>(x); }

    __device__ __forceinline__ T dot(const T (&u)[NC], const T (&v)[NC]) const
    {
        T s = (T)0;
#pragma unroll
        for (int i = 0; i < NC; i++) s = act[i] ? fma_t(u[i], v[i], s) : s;
        return rsum(s);
    }
    __device__ __forceinline__ T nrm2(const T (&u)[NC]) const { return (T)d_sqrt((double)dot(u, u)); }

    // k-vector in global memory -> registers (every copy loads it; inactive elements read as 0)
    __device__ __forceinline__ void start_point(const T* mrow, T (&x)[NC]) const { load_vec(mrow, x); }   // (lane_eval.hpp may have it prefetched)
    __device__ __forceinline__ void load_vec(const T* p, T (&x)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; i++) x[i] = act[i] ? p[elem[i]] : (T)0;
    }
    // registers -> global (copy 0 stores)
    __device__ __forceinline__ void store_vec(T* p, const T (&x)[NC]) const
    {
        if (jg == 0 && wid == 0 && member == 0) {
#pragma unroll
            for (int i = 0; i < NC; i++)
                if (act[i]) p[elem[i]] = x[i];
        }
    }

    // One batch of the gather: UU x 64 consecutive 16-byte slots starting at slot q0.  All UU global loads
    // are issued back to back (nothing consumes a result before the last one is in flight), then masked and
    // written to the tile.  Lanes past the end of the chunk load slot (0,0) again (an L1 hit) and store nothing.
    template <int UU> __device__ __forceinline__ void gather_batch(int q0, int Q, int& j, int& t)
    {
        SU v[UU];
        int dst[UU];
        bool last[UU];
#pragma unroll
        for (int u = 0; u < UU; u++) {
            const bool ok = q0 + u * WAVE + lane < Q;
            const int jr = ok ? j : 0, tr = ok ? t : 0;
            dst[u] = ok ? jr * s_stride + tr : -1;
            last[u] = tr == s_load - 1;
            const unsigned col = idxb[jr];
            v[u] = *(const SU*)(F + (size_t)col * (size_t)ldF + (size_t)(tr * SN));
            t += gdt; j += gdj;
            if (t >= s_load) { t -= s_load; j += 1; }
        }
#pragma unroll
        for (int u = 0; u < UU; u++) {
            SA w;
#pragma unroll
            for (int e = 0; e < SN; e++)  // the last slot of a factor row reads past its end: zero the excess
                w.v[e] = (e >= 1 && last[u] && e >= tail) ? (T)0 : v[u].v[e];
            if (dst[u] >= 0) tile[dst[u]] = w;
        }
    }

    // ---- chunk loading.  A chunk costs serialized memory round trips (~1.5-2 us each at load): its indices, then
    // its factor rows.  Two measures keep that at ONE exposed round trip per chunk: (i) up to 16 loads per lane
    // (16 KiB per wave) are put in flight before the first result is consumed; (ii) the indices / values of the
    // next chunk are fetched into registers while the current chunk is gathered and processed (meta_prefetch),
    // and dropped into LDS afterwards (meta_commit).
    __device__ __forceinline__ void meta_prefetch(unsigned c0, int cn)
    {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = lane + WAVE * u;
            if (j < cn) { m_idx[u] = ind[c0 + j]; m_x[u] = val[c0 + j]; }
        }
    }
    __device__ __forceinline__ void meta_commit(int cn)
    {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = lane + WAVE * u;
            if (j < cn) { idxb[j] = m_idx[u]; xb[j] = m_x[u]; }
        }
        wave_lds_fence();
    }
    __device__ __forceinline__ void meta_load(unsigned c0, int cn)
    {
        for (int j = lane; j < cn; j += WAVE) {
            idxb[j] = ind[c0 + j];
            xb[j] = val[c0 + j];
        }
        wave_lds_fence();
    }
    // factor rows of the chunk whose indices are in idxb -> tile
    __device__ __forceinline__ void gather_tile(int cn)
    {
        const int Q = cn * s_load;  // 16-byte slots to fetch
        int j = gj0, t = gt0;
        int q0 = 0;
        for (; q0 + 16 * WAVE <= Q; q0 += 16 * WAVE) gather_batch<16>(q0, Q, j, t);
        const int rem = Q - q0;
        if (rem > 8 * WAVE) gather_batch<16>(q0, Q, j, t);
        else if (rem > 4 * WAVE) gather_batch<8>(q0, Q, j, t);
        else if (rem > 2 * WAVE) gather_batch<4>(q0, Q, j, t);
        else if (rem > WAVE) gather_batch<2>(q0, Q, j, t);
        else if (rem > 0) gather_batch<1>(q0, Q, j, t);
        wave_lds_fence();
    }
    __device__ __forceinline__ void load_chunk(unsigned c0, int cn)
    {
        PMF_T0(tg);
        meta_load(c0, cn);
        gather_tile(cn);
        PMF_T1(0, tg);
    }
    // One chunk of a STREAMED row: uses prefetched indices when they are there, and prefetches those of the chunk
    // this wave will need next (the following one, or -- at the end of a pass -- the first one again, for the next
    // pass).  Call stream_chunk_done() after the chunk has been processed.
    __device__ __forceinline__ void stream_chunk_begin(unsigned c0, int cn, unsigned& nc0, int& ncn)
    {
        PMF_T0(tg);
        if (meta_c0 != c0) { meta_load(c0, cn); meta_c0 = c0; }
        nc0 = c0 + (unsigned)(NW * cap);
        if (nc0 >= nnz) nc0 = (unsigned)(wid * cap);   // wrap: the first chunk again, for the next pass over the row
        ncn = (int)((nnz - nc0 < (unsigned)cap) ? nnz - nc0 : (unsigned)cap);
        if (nc0 != c0 && cap <= 2 * WAVE) meta_prefetch(nc0, ncn);
        else ncn = 0;                                  // single-chunk row, or chunk too large for two registers per lane
        gather_tile(cn);
        PMF_T1(0, tg);
    }
    __device__ __forceinline__ void stream_chunk_done(unsigned nc0, int ncn)
    {
        if (ncn > 0) {
            meta_commit(ncn);
            meta_c0 = nc0;
        }
    }

    // ---- PF: the same chunk walk with the next chunk's tile requested one chunk ahead ------------------------------
    __device__ __forceinline__ unsigned next_chunk(unsigned c0) const
    {
        unsigned n = c0 + (unsigned)(NW * cap);
        if (n >= nnz) n = (unsigned)(wid * cap);       // wrap: the first chunk again, for the next pass over the row
        return n;
    }
    __device__ __forceinline__ int chunk_len(unsigned c0) const { return (int)((nnz - c0 < (unsigned)cap) ? nnz - c0 : (unsigned)cap); }
    // Makes chunk c0 current (tile + indices + values in LDS; a no-op when the previous visit prefetched it), then puts
    // the loads of the next chunk's factor rows in flight into `pre` and the indices of the chunk after that into
    // m_idx / m_x.  Returns the next chunk (== c0: nothing was requested).
    __device__ __forceinline__ unsigned visit_begin(unsigned c0, int cn, SU (&pre)[PRE])
    {
        PMF_T0(tg);
        if (meta_c0 != c0) { meta_load(c0, cn); meta_c0 = c0; }
        if (tile_c0 != c0) { gather_tile(cn); tile_c0 = c0; }
        const unsigned nc0 = next_chunk(c0);
        if (nc0 == c0) { PMF_T1(0, tg); return c0; }   // single-chunk share: the tile simply stays
        const int ncn = chunk_len(nc0);
        if (m_c0 == nc0) {                             // its indices were prefetched during the previous visit
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int j = lane + WAVE * u;
                if (j < ncn) { idxb2[j] = m_idx[u]; xb2[j] = m_x[u]; }
            }
        } else {
            for (int j = lane; j < ncn; j += WAVE) { idxb2[j] = ind[nc0 + j]; xb2[j] = val[nc0 + j]; }
        }
        wave_lds_fence();
        const unsigned nnc0 = next_chunk(nc0);
        meta_prefetch(nnc0, chunk_len(nnc0));
        m_c0 = nnc0;
        // factor rows of chunk nc0 -> registers (no wait)
        const int Q = ncn * s_load;
        int j = gj0, t = gt0;
#pragma unroll
        for (int u = 0; u < PRE; u++) {
            const bool ok = u * WAVE + lane < Q;
            const int jr = ok ? j : 0, tr = ok ? t : 0;
            const unsigned col = idxb2[jr];
            pre[u] = *(const SU*)(F + (size_t)col * (size_t)ldF + (size_t)(tr * SN));
            t += gdt; j += gdj;
            if (t >= s_load) { t -= s_load; j += 1; }
        }
        PMF_T1(0, tg);
        return nc0;
    }
    // After the current chunk has been processed: the prefetched chunk becomes the tile.
    __device__ __forceinline__ void visit_end(unsigned c0, unsigned nc0, const SU (&pre)[PRE])
    {
        if (nc0 == c0) return;
        wave_lds_fence();                              // every reader of the current tile / buffers is done
        const int Q = chunk_len(nc0) * s_load;
        int j = gj0, t = gt0;
#pragma unroll
        for (int u = 0; u < PRE; u++) {
            const bool ok = u * WAVE + lane < Q;
            SA w;
#pragma unroll
            for (int e = 0; e < SN; e++)  // the last slot of a factor row reads past its end: zero the excess
                w.v[e] = (e >= 1 && t == s_load - 1 && e >= tail) ? (T)0 : pre[u].v[e];
            if (ok) tile[j * s_stride + t] = w;
            t += gdt; j += gdj;
            if (t >= s_load) { t -= s_load; j += 1; }
        }
        T* tx = xb; xb = xb2; xb2 = tx;
        unsigned* ti = idxb; idxb = idxb2; idxb2 = ti;
        meta_c0 = nc0; tile_c0 = nc0;
        wave_lds_fence();
    }

    unsigned* ticket_word;  // NW > 1: one LDS word for the row-queue broadcast (last 16 bytes of the block)
    __device__ __forceinline__ unsigned* ticket_slot() const { return ticket_word; }
    __device__ __forceinline__ void begin_row(const unsigned* ind_, const T* val_, unsigned nnz_)
    {
        ind = ind_; val = val_; nnz = nnz_;
        meta_c0 = 0xffffffffu; tile_c0 = 0xffffffffu; m_c0 = 0xffffffffu;
        if (resident && nnz > 0) load_chunk(0, (int)nnz);
    }

    // Publish the point at which the next evaluations happen (copy 0 writes its slots; the excess of the
    // last slot is written as zero so that phase 1 can run over whole slots).
    __device__ __forceinline__ void set_point(const T (&x)[NC])
    {
        if (jg == 0) {
#pragma unroll
            for (int s = 0; s < NS; s++) {
                if (slot_on[s]) {
                    SA w;
#pragma unroll
                    for (int e = 0; e < SN; e++) w.v[e] = act[s * SN + e] ? x[s * SN + e] : (T)0;
                    avec[slotq[s]] = w;
                }
            }
        }
        wave_lds_fence();
    }

    // phase 1: this lane's nonzero (jb + lane) of the loaded chunk
    __device__ __forceinline__ T pred_lane(int jb, int cn) const
    {
        const int j = jb + lane;
        const SA* row = tile + (size_t)(j < cn ? j : cn - 1) * s_stride;
        // explicit 2-wide vectors so that the float build lowers to v_pk_fma_f32 on the register pairs the
        // ds_read_b128 results already sit in (left to itself the vectoriser pairs lanes (0,2),(1,3) and pays
        // three v_mov per packed FMA)
        typedef T V2 __attribute__((ext_vector_type(2)));
        constexpr int H = SN / 2;
        V2 p[H];
#pragma unroll
        for (int h = 0; h < H; h++) p[h] = (V2)(T)0;
        // Blocks of 8 slots: all 16 LDS reads of a block are in flight before the first FMA (at 1-2 waves per
        // SIMD nothing else hides LDS latency), then one masked block for the 0-7 slots left over.
        int t = 0;
        constexpr int PB = PMF_P1_BLOCK;
        for (; t + PB <= s_load; t += PB) {
            SA tv[PB], av[PB];
#pragma unroll
            for (int u = 0; u < PB; u++) { tv[u] = row[t + u]; av[u] = avec[t + u]; }
#pragma unroll
            for (int u = 0; u < PB; u++) {
#pragma unroll
                for (int h = 0; h < H; h++)
                    p[h] = __builtin_elementwise_fma((V2){ tv[u].v[2 * h], tv[u].v[2 * h + 1] },
                                                     (V2){ av[u].v[2 * h], av[u].v[2 * h + 1] }, p[h]);
            }
        }
        {
            const int rem = s_load - t;  // 0..PB-1, wave-uniform
            SA tv[PB - 1], av[PB - 1];
#pragma unroll
            for (int u = 0; u < PB - 1; u++) {
                if (u < rem) { tv[u] = row[t + u]; av[u] = avec[t + u]; }
            }
#pragma unroll
            for (int u = 0; u < PB - 1; u++) {
                if (u < rem) {
#pragma unroll
                    for (int h = 0; h < H; h++)
                        p[h] = __builtin_elementwise_fma((V2){ tv[u].v[2 * h], tv[u].v[2 * h + 1] },
                                                         (V2){ av[u].v[2 * h], av[u].v[2 * h + 1] }, p[h]);
                }
            }
        }
        if constexpr (SN == 4) return (p[0].x + p[0].y) + (p[1].x + p[1].y);
        else return p[0].x + p[0].y;
    }

    // coefficient buffer layout: the 64 nonzeros of the sub-chunk in flight, transposed so that group jg finds
    // the coefficients of ITS nonzeros (jg, jg + JG, ...) contiguously: coefb[jg * (64 / JG) + step]
    __device__ __forceinline__ int coef_slot(int j) const { return (j % JG) * (WAVE / JG) + j / JG; }

    // UB steps of phase 2 starting at step `it`.  MASKED = false: every nonzero touched exists, so the UB tile
    // rows are at constant strides from one base address; MASKED = true (tail): rows past the end are clamped to
    // the last valid row (their coefficient is zero, the tile beyond the chunk may hold anything).
    template <bool UNIT, int UB, bool MASKED> __device__ __forceinline__ void accumulate_block(const SA* base, int it, int cnt,
                                                                                               T (&part)[NC]) const
    {
        SA tv[UB][NS];
        T cj[UB];
        const T* cf = coefb + jg * (WAVE / JG) + it;
        const SA* row0 = base + (size_t)(it * JG + jg) * s_stride;
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int j = (it + u) * JG + jg;
            const SA* row = row0 + (size_t)(u * JG) * s_stride;
            if constexpr (MASKED) row = base + (size_t)(j < cnt ? j : cnt - 1) * s_stride;
            if constexpr (UNIT) cj[u] = (!MASKED || j < cnt) ? (T)1 : (T)0;
            else cj[u] = cf[u];
#pragma unroll
            for (int s = 0; s < NS; s++) tv[u][s] = row[slotq[s]];
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
#pragma unroll
            for (int s = 0; s < NS; s++) {
#pragma unroll
                for (int e = 0; e < SN; e++) part[s * SN + e] = fma_t(cj[u], tv[u][s].v[e], part[s * SN + e]);
            }
        }
    }
    // phase 2 for the cnt (<= 64) nonzeros starting at tile row jb whose coefficients are in coefb:
    // group jg accumulates nonzeros jg, jg + JG, ...  (UNIT: all coefficients are 1, coefb is not read)
    template <bool UNIT> __device__ __forceinline__ void accumulate(int jb, int cnt, T (&part)[NC]) const
    {
        const SA* base = tile + (size_t)jb * s_stride;
        const int full = cnt / JG;              // steps in which every group has a nonzero (wave-uniform)
        const int steps = (cnt + JG - 1) / JG;
        int it = 0;
        if constexpr (PMF_P2_BLOCK >= 16) {
            for (; it + 16 <= full; it += 16) accumulate_block<UNIT, 16, false>(base, it, cnt, part);
            if (it + 8 <= full) { accumulate_block<UNIT, 8, false>(base, it, cnt, part); it += 8; }
        } else {
            for (; it + 8 <= full; it += 8) accumulate_block<UNIT, 8, false>(base, it, cnt, part);
        }
        if (it + 4 <= full) { accumulate_block<UNIT, 4, false>(base, it, cnt, part); it += 4; }
        if (it < steps) accumulate_block<UNIT, 4, true>(base, it, cnt, part);   // <= 3 full steps + a partial one
    }

    // combine the JG per-group partial sums (fixed order, identical in every copy) and add them to acc
    __device__ __forceinline__ void combine_groups(T (&part)[NC], T (&acc)[NC]) const
    {
        if (JG >= 2) {
            if (JG == 4) {
#pragma unroll
                for (int i = 0; i < NC; i++) part[i] += __shfl_xor(part[i], 16);
            }
#pragma unroll
            for (int i = 0; i < NC; i++) part[i] += __shfl_xor(part[i], 32);
        }
#pragma unroll
        for (int i = 0; i < NC; i++) acc[i] += part[i];
    }

    // NW > 1: add up the NW waves' partial results (fixed order; every wave ends with the same bits)
    __device__ __forceinline__ void combine_waves(T (&tot)[NC], double& lsum)
    {
        if constexpr (NW > 1) {
            // Alternating scratch sets: a wave may run ahead into the NEXT combine (other set) while a slow wave still
            // reads this one, but cannot reach the one after that (this set again) before everybody has passed the next
            // barrier -- one barrier per evaluation.  Whole 16-byte slots go through LDS, all reads are put in flight
            // before the first add.
            double* red_l = (double*)(red_base + red_sel * red_bytes);
            SA* red_slots = (SA*)(red_base + red_sel * red_bytes + 16 * ((NW * sizeof(double) + 15) / 16));
            red_sel ^= 1;
            if (jg == 0) {
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    if (slot_on[s]) {
                        SA v;
#pragma unroll
                        for (int e = 0; e < SN; e++) v.v[e] = act[s * SN + e] ? tot[s * SN + e] : (T)0;
                        red_slots[wid * s_load + slotq[s]] = v;
                    }
                }
            }
            if (lane == 0) red_l[wid] = lsum;
            __syncthreads();
            SA part[NW][NS];
            double lp[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                lp[w] = red_l[w];
#pragma unroll
                for (int s = 0; s < NS; s++) part[w][s] = red_slots[w * s_load + slotq[s]];   // slotq is clamped into the row
            }
            lsum = 0.0;
#pragma unroll
            for (int i = 0; i < NC; i++) tot[i] = (T)0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                lsum += lp[w];
#pragma unroll
                for (int s = 0; s < NS; s++) {
#pragma unroll
                    for (int e = 0; e < SN; e++) tot[s * SN + e] += act[s * SN + e] ? part[w][s].v[e] : (T)0;
                }
            }
            if constexpr (TM) team_exchange(tot, lsum);
        }
    }
    // TM, every wave of the member, after the member's own waves have been added up (tot / lsum identical in all of them): the member's
    // sums cross the team.  Wave 0 publishes them as tagged granules { 32 data bits | exchange number } (one store each: data and tag
    // arrive together; two alternating sets, as a member can be at most one exchange ahead of the slowest), collects the other members'
    // and adds everything in member order; the totals reach the other waves through the cross-wave scratch, between two barriers.
    __device__ __forceinline__ void team_exchange(T (&tot)[NC], double& lsum)
    {
        tm_seq++;
        double* red_l = (double*)(red_base + red_sel * red_bytes);
        SA* red_slots = (SA*)(red_base + red_sel * red_bytes + 16 * ((NW * sizeof(double) + 15) / 16));   // (the set the NEXT combine writes: nobody reads it now)
        if (wid == 0) {
            int idx[NC];
            bool valid[NC];
            double v[NC];
#pragma unroll
            for (int s = 0; s < NS; s++) {
#pragma unroll
                for (int e = 0; e < SN; e++) {
                    idx[s * SN + e] = slotq[s] * SN + e;
                    valid[s * SN + e] = jg == 0 && slot_on[s];
                    v[s * SN + e] = (double)(act[s * SN + e] ? tot[s * SN + e] : (T)0);
                }
            }
            double lt = lsum;
            (void)team_sum<NC>(tm_words, tm_M, member, tm_seq, tm_err, tm_spin, lane, idx, valid, v, lt);
            if (jg == 0) {
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    if (slot_on[s]) {
                        SA w;
#pragma unroll
                        for (int e = 0; e < SN; e++) w.v[e] = act[s * SN + e] ? (T)v[s * SN + e] : (T)0;
                        red_slots[slotq[s]] = w;
                    }
                }
            }
            if (lane == 0) red_l[0] = lt;
        }
        __syncthreads();
        lsum = red_l[0];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const SA v = red_slots[slotq[s]];
#pragma unroll
            for (int e = 0; e < SN; e++) tot[s * SN + e] = act[s * SN + e] ? v.v[e] : (T)0;
        }
        __syncthreads();   // (the next combine writes this set)
    }

    // At the point last published with set_point:
    //   WANT_F : returns lsum = sum_j x_j log(pred_j)   (log and the sum in double, as the reference's
    //            `lsum += X[ix] * log(dot)` is a double expression even in its float build)
    //   WANT_G : acc_c += sum_j (sgn x_j / pred_j) F[ind_j, c]
    //   store  : if not null, pred_j is also written to store[j] for every nonzero j of the row
    template <bool WANT_F, bool WANT_G> __device__ __forceinline__ double eval(T sgn, T (&acc)[NC], T* store = nullptr)
    {
        n_eval++;
        double lpart = 0.0;
        T part[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) part[i] = (T)0;
        for (unsigned c0 = (unsigned)(wid * cap); c0 < nnz; c0 += (unsigned)(NW * cap)) {
            const int cn = (int)((nnz - c0 < (unsigned)cap) ? nnz - c0 : (unsigned)cap);
            unsigned nc0 = 0;
            int ncn = 0;
            SU pre[PRE];
            if (!resident) {
                if (PF && pf_on) nc0 = visit_begin(c0, cn, pre);
                else stream_chunk_begin(c0, cn, nc0, ncn);
            }
            for (int jb = 0; jb < cn; jb += WAVE) {
                PMF_T0(t1);
                const T pred = pred_lane(jb, cn);
                PMF_T1(1, t1);
                PMF_T0(t2);
                const bool on = jb + lane < cn;
                const T xj = xb[on ? jb + lane : 0];
                if (store != nullptr && on) store[c0 + jb + lane] = pred;
                if constexpr (WANT_F) lpart += on ? (double)xj * d_log((double)pred) : 0.0;
                if constexpr (WANT_G) {
                    wave_lds_fence();  // previous sub-chunk's readers of coefb are done
                    coefb[coef_slot(lane)] = on ? sgn * xj / pred : (T)0;
                    wave_lds_fence();
                    PMF_T1(2, t2);
                    PMF_T0(t3);
                    accumulate<false>(jb, cn - jb < WAVE ? cn - jb : WAVE, part);
                    PMF_T1(3, t3);
                } else {
                    PMF_T1(2, t2);   // function evaluations: the log is the "coefficient" step
                }
            }
            if (!resident) {
                if (PF && pf_on) visit_end(c0, nc0, pre);
                else stream_chunk_done(nc0, ncn);
            }
        }
        PMF_T0(t4);
        if constexpr (NW > 1) {
            T tot[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) tot[i] = (T)0;
            if constexpr (WANT_G) combine_groups(part, tot);
            double lsum = 0.0;
            if constexpr (WANT_F) lsum = wave_sum(lpart);
            combine_waves(tot, lsum);
            if constexpr (WANT_G) {
#pragma unroll
                for (int i = 0; i < NC; i++) acc[i] += tot[i];
            }
            return lsum;
        }
        if constexpr (WANT_G) combine_groups(part, acc);
        PMF_T1(4, t4);
        if (store != nullptr) wave_lds_fence();
        if constexpr (WANT_F) return wave_sum(lpart);
        else return 0.0;
    }

    // sum_j x_j log(p_j + alpha q_j) from the cached predictions: no access to the tile or to F at all.
    // This is the evaluation the reference's authors describe as the faster alternative for line searches
    // (ref: src/poismf.c:191-193, src/nonnegcg.c:291-294).
    // `trusted` comes back false when some p_j + alpha q_j cancels to (almost) nothing (see RegEval::logsum_cached): the
    // caller then evaluates that trial directly.
    static constexpr bool PARKS = false;
    static constexpr bool CACHED_GRAD = false;
    static constexpr int LS_BATCH = 1;
    __device__ __forceinline__ void logsum_cached_batch(T alpha, T, double (&ls)[1], bool (&trusted)[1]) const { ls[0] = logsum_cached(alpha, trusted[0]); }
    __device__ __forceinline__ double logsum_cached(T alpha, bool& trusted) const
    {
        double lpart = 0.0;
        bool bad = false;
        for (unsigned j = lane; j < nnz; j += WAVE) {
            const T pj = pbuf[j];
            const T pred = fma_t(alpha, qbuf[j], pj);
            bad = bad || !(pred > pj * (T)1e-4);
            lpart += (double)val[j] * d_log((double)pred);
        }
        trusted = __builtin_amdgcn_ballot_w64(bad) == 0;
        return wave_sum(lpart);
    }
    // after an accepted step x <- x + alpha d the cached T.x moves along with it
    __device__ __forceinline__ void advance_cached(T alpha)
    {
        for (unsigned j = lane; j < nnz; j += WAVE) pbuf[j] = fma_t(alpha, qbuf[j], pbuf[j]);
        wave_lds_fence();
    }

    // acc_c += sum_j F[ind_j, c]   (the gather pass of adjustment_Bsum, ref: src/poismf.c:108-110,
    // served from the tile instead of a second trip to memory)
    __device__ __forceinline__ void tile_colsum(T (&acc)[NC])
    {
        T part[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) part[i] = (T)0;
        for (unsigned c0 = (unsigned)(wid * cap); c0 < nnz; c0 += (unsigned)(NW * cap)) {
            const int cn = (int)((nnz - c0 < (unsigned)cap) ? nnz - c0 : (unsigned)cap);
            unsigned nc0 = 0;
            int ncn = 0;
            SU pre[PRE];
            if (!resident) {
                if (PF && pf_on) nc0 = visit_begin(c0, cn, pre);
                else stream_chunk_begin(c0, cn, nc0, ncn);
            }
            for (int jb = 0; jb < cn; jb += WAVE) accumulate<true>(jb, cn - jb < WAVE ? cn - jb : WAVE, part);
            if (!resident) {
                if (PF && pf_on) visit_end(c0, nc0, pre);
                else stream_chunk_done(nc0, ncn);
            }
        }
        if constexpr (NW > 1) {
            T tot[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) tot[i] = (T)0;
            combine_groups(part, tot);
            double unused = 0.0;
            combine_waves(tot, unused);
#pragma unroll
            for (int i = 0; i < NC; i++) acc[i] += tot[i];
            return;
        }
        combine_groups(part, acc);
    }
};

}  // namespace pmf
