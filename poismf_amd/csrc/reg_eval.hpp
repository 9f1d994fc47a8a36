// reg_eval.hpp -- the per-row evaluation engine for SHORT rows: the gathered tile lives in REGISTERS.
//
// Same contract as RowEval (row_eval.hpp): one wavefront owns one output row, the factor rows F[ind_j] named by the
// row's nonzeros are fetched once and every inner pass of the solver runs on chip.  Here "on chip" is the vector
// register file instead of LDS: for rows of up to JG*S nonzeros whose factor rows fit 16 slots of 16 bytes (fp32
// k <= 64, fp64 k <= 32) the whole tile is S*NS slots per lane (k = 50 fp32, 100 nonzeros: ~100 VGPRs), there is no
// LDS traffic for the tile at all, and the waves per CU are set by registers (8-16) instead of by a 20-40 KiB LDS
// tile (4-7).  Both phases of an evaluation work on ONE layout, the slot layout the solvers keep their k-vectors in:
//
//   lane = (jg, g): g = lane % G, jg = lane / G; a group of G (8 or 16) lanes holds one factor row, lane g its
//   16-byte slots g, g + G, .. (NS per lane); the wave holds JG = 64 / G groups;
//   step s (0 <= s < S) handles the JG nonzeros j = JG s + jg, one per group; t[s][n] is slot g + G n of F[ind_j].
//
//   dots     pred_j = F[ind_j] . a : every lane multiplies its slots of t[s] with its slots of `a` (packed FMAs), and
//            the G lane-partials of a step are summed over the group.  G steps are reduced TOGETHER by a transposing
//            butterfly (log2 G levels of select-and-add pairs: G - 1 folds instead of G log2 G DPP adds) that leaves
//            the finished pred of step G b + g in lane g -- 64 distinct nonzeros in 64 lanes, so the division
//            x_j / pred_j (and the double-precision log) is done exactly once per nonzero;
//   axpy     acc += coef_j * t[s] for s in step order: coef of step u comes from lane u of the group by ds_swizzle
//            (the LDS crossbar, no LDS memory), packed FMAs; the JG groups' partial sums are combined once per
//            evaluation.  Nonzero -> group assignment and summation order are those of RowEval's phase 2.
//
// This is the reference's per-nonzero ddot + daxpy (ref: src/poismf.c:126-133 calc_grad_pgd, :194-208
// calc_fun_single, :210-240 calc_grad_single[_w], :242-273 calc_fun_and_grad).
#pragma once
#include <type_traits>

#include "row_eval.hpp"

namespace pmf {

template <int I, int N, class Fn> __device__ __forceinline__ void static_for(Fn&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// value of lane U of the caller's group of G lanes (ds_swizzle, bit-mask mode inside each half-wave:
// src = (lane & ~(G - 1)) | U)
template <int G, int U> __device__ __forceinline__ int group_bcast_i32(int v)
{
    return __builtin_amdgcn_ds_swizzle(v, (0x1f & ~(G - 1)) | (U << 5));
}
template <int G, int U> __device__ __forceinline__ unsigned group_bcast(unsigned v) { return (unsigned)group_bcast_i32<G, U>((int)v); }
template <int G, int U> __device__ __forceinline__ float group_bcast(float v)
{
    return __builtin_bit_cast(float, group_bcast_i32<G, U>(__builtin_bit_cast(int, v)));
}
template <int G, int U> __device__ __forceinline__ double group_bcast(double v)
{
    // sixteen lanes are one DPP row: v_mov_b64_dpp row_newbcast (the one DPP control 64-bit operands take) instead of two trips
    // through the LDS crossbar -- the fp64 kernels run one wave per SIMD and wait every such trip out
    if constexpr (G == 16) return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + U, 0xf, 0xf, true);
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)group_bcast_i32<G, U>((int)(unsigned)b);
    const unsigned hi = (unsigned)group_bcast_i32<G, U>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// One level of the transposing butterfly.  Lanes whose class bit is clear keep collecting `pa`, lanes whose bit is
// set keep collecting `pb`; the partner lane (CTRL: an involution that flips exactly that class bit) supplies its
// copy of the register this lane collects.
template <int CTRL, class T> __device__ __forceinline__ T fold_pair(bool cls, T pa, T pb)
{
    const T own = cls ? pb : pa;
    const T send = cls ? pa : pb;
    return own + dpp_mov<CTRL>(send);
}
template <int CTRL, class T> __device__ __forceinline__ T fold_one(T pa) { return pa + dpp_mov<CTRL>(pa); }

// NIN (<= 2 H) inputs -> H outputs (output i collects input i in lanes of class 0 and input i + H in lanes of class 1)
template <int CTRL, int H, int NIN, class T> __device__ __forceinline__ void fold_level(bool cls, const T* in, T* out)
{
#pragma unroll
    for (int i = 0; i < H; i++) {
        if (i + H < NIN) out[i] = fold_pair<CTRL>(cls, in[i], in[i + H]);
        else if (i < NIN) out[i] = fold_one<CTRL>(in[i]);
        else out[i] = (T)0;
    }
}

// num / den for the gradient coefficients.  fp32: reciprocal + one Newton step on the quotient + v_div_fixup (exact
// handling of 0, inf and NaN operands) -- 5 instructions instead of the 14 of the IEEE sequence, within 1 ulp of it
// (the tolerance of every fp32 comparison in tests/ is >= 1e-5).  fp64: IEEE division.
__device__ __forceinline__ float coef_div(float num, float den)
{
    const float r = __builtin_amdgcn_rcpf(den);
    float q = num * r;
    const float e = __builtin_fmaf(-den, q, num);
    q = __builtin_fmaf(e, r, q);
    return __builtin_amdgcn_div_fixupf(q, den, num);
}
__device__ __forceinline__ double coef_div(double num, double den) { return num / den; }

// The two upper levels of the butterfly on floats, all folds of a level in one block: the class of a lane is a set of
// DPP banks there (lane ^ 8: banks {0,1} | {2,3}; lane ^ 7: banks {0,2} | {1,3}), and a DPP add leaves the lanes of
// masked-off banks untouched, so a fold is TWO `v_add_f32_dpp` into the same register with complementary bank masks
// instead of two selects and an add.  (Not expressible through the update_dpp builtin: its masked-off lanes take `old`
// before the add.)  s_nop 1: a DPP source written by the previous VALU instruction needs two wait states, and the
// hazard recogniser does not look inside inline asm.
#define PMF_FOLD2(CTRL, M0, M1, Q, A, B) \
    "v_add_f32_dpp " Q ", " A ", " A " " CTRL " row_mask:0xf bank_mask:" M0 "\n\t" \
    "v_add_f32_dpp " Q ", " B ", " B " " CTRL " row_mask:0xf bank_mask:" M1 "\n\t"
__device__ __forceinline__ void fold16_banked(const float (&p)[16], float (&q)[8])
{
    asm("s_nop 1\n\t"
        PMF_FOLD2("row_ror:8", "0x3", "0xc", "%0", "%8", "%16")
        PMF_FOLD2("row_ror:8", "0x3", "0xc", "%1", "%9", "%17")
        PMF_FOLD2("row_ror:8", "0x3", "0xc", "%2", "%10", "%18")
        PMF_FOLD2("row_ror:8", "0x3", "0xc", "%3", "%11", "%19")
        PMF_FOLD2("row_ror:8", "0x3", "0xc", "%4", "%12", "%20")
        PMF_FOLD2("row_ror:8", "0x3", "0xc", "%5", "%13", "%21")
        PMF_FOLD2("row_ror:8", "0x3", "0xc", "%6", "%14", "%22")
        PMF_FOLD2("row_ror:8", "0x3", "0xc", "%7", "%15", "%23")
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]),
          "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]));
}
__device__ __forceinline__ void fold8_banked(const float (&q)[8], float (&r)[4])
{
    asm("s_nop 1\n\t"
        PMF_FOLD2("row_half_mirror", "0x5", "0xa", "%0", "%4", "%8")
        PMF_FOLD2("row_half_mirror", "0x5", "0xa", "%1", "%5", "%9")
        PMF_FOLD2("row_half_mirror", "0x5", "0xa", "%2", "%6", "%10")
        PMF_FOLD2("row_half_mirror", "0x5", "0xa", "%3", "%7", "%11")
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
        : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]));
}

// NW_ > 1: a workgroup of NW_ wavefronts shares ONE row, for rows longer than one wave's registers hold (k = 50 fp32,
// NW_ = 8: up to 1280 nonzeros stay on chip).  Wave w keeps nonzeros [w C, (w + 1) C) of the row in its tile, C =
// ceil(nnz / NW_) rounded up to a whole step; every wave keeps its own copy of the solver state and runs the same
// wave-uniform control flow; the NW_ partial gradients / log-likelihood sums of an evaluation are added through LDS in
// wave order behind a workgroup barrier, so all copies stay bit-identical (the scheme of RowEval's long-row path).
// tagged 8-byte granules between CUs (MI355X_MICROARCH "inter-workgroup visibility": 8-byte agent-scope atomics on both sides
// are coherent across XCDs without fences; a granule is written by ONE store, so data and tag arrive together)
__device__ __forceinline__ void gran_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long gran_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// M_ > 1: a TEAM of M_ workgroups (one per CU) shares one row, for rows whose tile fits no single CU's registers (k = 50 fp64,
// 1000 nonzeros: 400 KB).  Member m keeps nonzeros [m NW_ C, (m + 1) NW_ C) spread over its NW_ waves as above; every wave of
// every member keeps its own copy of the solver state and runs the same control flow.  An evaluation first adds up the
// member's waves through LDS, then wave 0 publishes the member's sums as tagged granules, collects the other members' and
// adds them in member order (every member ends with the same bits), and hands the totals to its workgroup through LDS.
// Tags count the team's exchanges; two slots per member alternate (a member can be at most one exchange ahead of the
// slowest: it needs everybody's granules of exchange n before it can publish n + 1, and everybody publishes n only after
// reading n - 1).  Teams form in arrival order per XCD (poismf_hip.hip, half_sweep_team_kernel), so a team's members are
// resident by construction and a launch makes progress with any two workgroups of an XCD on the chip.
template <class T, int S, int G_ = 16, int NS_ = 1, int NW_ = 1, int M_ = 1> struct RegEval {
    using SA = typename Slot<T>::A;
    using SU = typename Slot<T>::U;
    static constexpr int SN = Slot<T>::N;
    static constexpr int G = G_, NS = NS_;
    static constexpr int JG = WAVE / G;           // nonzeros per step
    static constexpr int NC = NS * SN;            // elements per lane
    static constexpr int NB = (S + G - 1) / G;    // batches of G steps = 64 nonzeros
    static constexpr int NW = NW_;
    static constexpr bool PIPELINED = true;       // sweep_rows prefetches the next row's indices during the solver
    static constexpr int PIPE_MW = 1;             // (lane_eval.hpp: whether multi-wave rows take the pipeline depends on the solver)
    static constexpr bool FUSED_SUMS = false;     // (lane_eval.hpp reduces the solvers' groups of dot products together)
    static constexpr bool PREFETCH = false;       // (lane_eval.hpp can request the next row's tile while this one is solved)
    static constexpr int KP = G * NS * SN;        // elements of a (padded) k-vector in the cross-wave scratch
    static constexpr int M = M_;
    // one set of cross-wave scratch: NW partial k-vectors and NW scalars (teams: NW x TEAM_SC scalars)
    static constexpr int RED_BYTES = NW_ * KP * (int)sizeof(T) + (M_ > 1 ? NW_ * 8 * TEAM_SC : 16 * ((NW_ * 8 + 15) / 16));
    static constexpr int TEAM_BYTES = M_ > 1 ? KP * (int)sizeof(T) + 8 * TEAM_SC : 0;           // team totals: a k-vector and the scalars
    // PARKS: six k-vectors per wave wait in LDS during the passes of cg_row_cached (solvers.hpp)
    static constexpr bool PARKS = sizeof(T) == 8 && M_ > 1;
    static constexpr int PARK_SLOTS = 6;
    static constexpr int PARK_BYTES = PARKS ? NW_ * PARK_SLOTS * KP * (int)sizeof(T) : 0;
    static constexpr int PARK_OFFSET = (NW_ > 1 ? 2 * RED_BYTES + 16 : 0) + TEAM_BYTES;
    static constexpr int SMEM_BYTES = PARK_OFFSET + PARK_BYTES;
    static_assert(M_ == 1 || (sizeof(T) == 8 && NW_ > 1 && NS_ * Slot<T>::N <= WAVE / G_), "teams: doubles, one element per group to publish");
    static_assert(G == 8 || G == 16, "a factor row is held by 8 or 16 lanes");

    SA t[S][NS];    // the tile
    T a[NC];        // current point (this lane's slots)
    T xr[NB];       // x_j of the nonzero whose pred this lane finishes in batch b: j = 64 b + JG g + jg
    unsigned idx_n[NB];  // column indices of the row whose tile is requested next, same layout (fetch_meta -> gather)
    // launch constants
    const T* F;
    unsigned zero_row;
    int k, ldF, s_load;
    int lane, g, jg, wid;
    int jlane;      // JG g + jg
    bool cls8, cls4, cls2, cls1;
    struct ElemOf {   // factor dimension held in element i of this lane (computed, not kept: registers are what sets the waves per SIMD)
        int g;
        __device__ __forceinline__ int operator[](int i) const { return (g + G * (i / SN)) * SN + i % SN; }
    } elem;
    bool act[NC];
    bool slot_on[NS];
    unsigned nnz;   // nonzeros of the row held by THIS wave
    unsigned n_eval; // passes over the tile since the caller last reset it (wave-uniform; reporting only)
#ifdef PMF_PROBE
    unsigned* probe = nullptr;
    unsigned long long probe_wait = 0;   // teams: cycles wave 0 spent polling the other members' granules
    unsigned long long probe_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // teams: cycles per phase, summed over the kernel (probe_team.py)
#endif
    unsigned char* red_base;  // NW > 1: two sets of { [NW][KP] partial gradients, [NW] partial log-likelihood sums }
    int red_sel;              // the set the next combine_waves uses (alternating sets: one barrier per evaluation)
    unsigned* ticket_word;
    T* park_base;                     // PARKS: this wave's PARK_SLOTS k-vectors in LDS
    // team state (M > 1)
    unsigned long long* team_words;   // this team's words of HalfArgs::team_buf
    unsigned* team_err;               // != 0: some exchange of this launch timed out, give up
    unsigned team_spin = TEAM_SPIN_LIMIT;   // polls before giving up
    T* team_tot;                      // LDS: the team's totals
    int member;                       // 0 .. M - 1 (0 when M == 1)
    unsigned xseq;                    // exchanges so far
    // Cached line search (solvers.hpp, cg_row_cached): the predictions p_j = F_j . x and q_j = F_j . d of the nonzero each
    // lane finishes stay in registers (pv / qv, one per batch); pbuf / qbuf are only tags that tell eval() which of the
    // two a pass is to keep.  First used by the fp64 single-wave kernels (pq_cap > 0): one wave per SIMD, where an Armijo trial
    // as two logs instead of a pass over the tile is what shortens the row (C3 CG fp64 A half: 37.3 -> see DESIGN.md); the
    // teams and, since the cache became a compile-time property of an instance, the fp32 kernels take it too.
    int pq_cap;
    T* pbuf;
    T* qbuf;
    static constexpr bool CACHED = (sizeof(T) == 8 && (NW_ == 1 || M_ > 1)) || sizeof(T) == 4;   // compile-time: no trace of the cache in the other instances
    static constexpr bool MAY_CACHE = CACHED;
    static constexpr bool CACHED_GRAD = CACHED;   // cg_row_cached may take gradients from the cached predictions
    T pv[CACHED ? NB : 1], qv[CACHED ? NB : 1];

    __device__ __forceinline__ void init(const TileGeom& geo, const T* F_, unsigned char* smem)
    {
        lane = lane_id();
        red_base = smem;
        red_sel = 0;
        ticket_word = (unsigned*)(smem + 2 * RED_BYTES);
        team_tot = (T*)(smem + 2 * RED_BYTES + 16);
        park_base = (T*)(smem + PARK_OFFSET) + (NW > 1 ? (int)(threadIdx.x / WAVE) : 0) * PARK_SLOTS * KP;
        member = 0; xseq = 0; team_words = nullptr; team_err = nullptr;
        F = F_;
        k = geo.k; ldF = geo.ldF; s_load = geo.s_load; zero_row = geo.zero_row;
        g = lane & (G - 1); elem.g = g; jg = lane / G; wid = NW > 1 ? (int)(threadIdx.x / WAVE) : 0;
        jlane = JG * g + jg;
        cls8 = (lane & 8) != 0; cls4 = (lane & 4) != 0; cls2 = (lane & 2) != 0; cls1 = (lane & 1) != 0;
#pragma unroll
        for (int n = 0; n < NS; n++) {
            const int q = g + G * n;
            slot_on[n] = q < s_load;
#pragma unroll
            for (int e = 0; e < SN; e++) {
                act[n * SN + e] = q * SN + e < k;
            }
        }
        pq_cap = CACHED ? 0x7fffffff : 0;
        pbuf = (T*)(size_t)16; qbuf = (T*)(size_t)32;   // tags, never dereferenced
    }

    // ---- k-length vector helpers (the slot layout of RowEval, with G = 8 as well) ----------------------
    template <class Op, class V> __device__ __forceinline__ V reduce(V x) const
    {
        x = Op::f(x, dpp_mov<0xB1>(x));
        x = Op::f(x, dpp_mov<0x4E>(x));
        x = Op::f(x, dpp_mov<0x141>(x));
        if constexpr (G == 16) x = Op::f(x, dpp_mov<0x140>(x));
        return uniform(x);
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
    // k-vector in global memory <-> registers, one 16-byte access per slot (the vectors this engine touches -- factor
    // rows, the column-sum vector -- all have >= 16 bytes of slack behind them, so the last, partly filled slot may be
    // read whole; it is written element by element)
    __device__ __forceinline__ void start_point(const T* mrow, T (&x)[NC]) const { load_vec(mrow, x); }   // (lane_eval.hpp may have it prefetched)
    __device__ __forceinline__ void load_vec(const T* p, T (&x)[NC]) const
    {
#pragma unroll
        for (int n = 0; n < NS; n++) {
            SU v;
#pragma unroll
            for (int e = 0; e < SN; e++) v.v[e] = (T)0;
            if (slot_on[n]) v = *(const SU*)(p + (g + G * n) * SN);
#pragma unroll
            for (int e = 0; e < SN; e++) x[n * SN + e] = act[n * SN + e] ? v.v[e] : (T)0;
        }
    }
    __device__ __forceinline__ void store_vec(T* p, const T (&x)[NC]) const
    {
        if (jg == 0 && wid == 0 && member == 0) {
#pragma unroll
            for (int n = 0; n < NS; n++) {
                if (act[n * SN + SN - 1]) {          // whole slot inside the row
                    SU v;
#pragma unroll
                    for (int e = 0; e < SN; e++) v.v[e] = x[n * SN + e];
                    *(SU*)(p + (g + G * n) * SN) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < SN; e++)
                        if (act[n * SN + e]) p[elem[n * SN + e]] = x[n * SN + e];
                }
            }
        }
    }

    // a k-vector to / from this wave's LDS slot (the four groups hold the same values: one writes, all read)
    __device__ __forceinline__ void park(int slot, const T (&x)[NC])
    {
        if (jg == 0) {
#pragma unroll
            for (int n = 0; n < NS; n++) {
                SA v;
#pragma unroll
                for (int e = 0; e < SN; e++) v.v[e] = x[n * SN + e];
                *(SA*)(park_base + slot * KP + (g + G * n) * SN) = v;
            }
        }
    }
    __device__ __forceinline__ void unpark(int slot, T (&x)[NC])
    {
#ifdef PMF_PROBE
        const unsigned long long t_up0 = __builtin_amdgcn_s_memtime();
        struct UpTimer { unsigned long long& acc; unsigned long long t0; __device__ ~UpTimer() { acc += __builtin_amdgcn_s_memtime() - t0; } } up_timer{ probe_acc[3], t_up0 };
#endif
        wave_lds_fence();
#pragma unroll
        for (int n = 0; n < NS; n++) {
            const SA v = *(const SA*)(park_base + slot * KP + (g + G * n) * SN);
#pragma unroll
            for (int e = 0; e < SN; e++) x[n * SN + e] = v.v[e];
        }
    }

    // ---- gather: indices / values coalesced in the "finishing lane" layout, then NS 16-byte loads per step ----
    // All loads of a row are issued back to back, unconditionally: steps past the end of the row and lanes whose
    // slot does not exist fetch from row `zero_row` = dimF, an all-zero row the session keeps behind the factor, and
    // the gathered copy of the factor has rows of whole 16-byte slots ending in zeros (the session pads it when
    // k * sizeof(T) is not a multiple of 16), so nothing needs masking afterwards.
    // Split in two so that sweep_rows can run fetch_meta for the NEXT row while the solver works on the current one.
    // this wave's share [c0, c0 + mine) of a row of nnz_row nonzeros (NW = 1: the whole row)
    __device__ __forceinline__ void my_share(unsigned nnz_row, unsigned& c0, unsigned& mine) const
    {
        if constexpr (NW > 1) {
            const unsigned C = ((nnz_row + NW * M - 1) / (NW * M) + JG - 1) / JG * JG;
            c0 = (unsigned)(member * NW + wid) * C;
            mine = c0 < nnz_row ? (nnz_row - c0 < C ? nnz_row - c0 : C) : 0u;
        } else {
            c0 = 0u; mine = nnz_row;
        }
    }
    // ind / val: the row's first nonzero; nnz_row: its length
    __device__ __forceinline__ void fetch_meta(const unsigned* ind, unsigned nnz_row)
    {
        unsigned c0, mine;
        my_share(nnz_row, c0, mine);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const unsigned j = (unsigned)(64 * b + jlane);
            idx_n[b] = j < mine ? ind[c0 + j] : zero_row;   // steps past the end of the row fetch the all-zero row behind F
        }
    }
    __device__ __forceinline__ void begin_row(const unsigned* ind, const T* val, unsigned nnz_row)
    {
        fetch_meta(ind, nnz_row);
        gather(val, nnz_row);
    }
    __device__ __forceinline__ unsigned* ticket_slot() const { return ticket_word; }

    // M > 1, every wave of the member, after the member's own waves have been added up (tot / sc identical in all of them):
    // the k-vector (vec) and TEAM_SC scalars cross the team.  (Every wave collecting the other members' granules itself -- 8
    // loads per lane instead of wave 0's two, no second barrier and no LDS hand-over: C3 B half 40.9 -> 43.8 ms.)
    __device__ __forceinline__ void team_exchange(T (&tot)[NC], double (&sc)[TEAM_SC], bool vec)
    {
        if constexpr (M > 1) {
            xseq++;
            double* team_l = (double*)(team_tot + KP);
            if (wid == 0) {
                const unsigned long long tag = (unsigned long long)xseq << 32;
                unsigned long long* slots = team_words + 2 + TEAM_M_MAX + (size_t)(xseq & 1u) * TEAM_M_MAX * TEAM_GRAN;
                // lane (jg, g) carries element jg of its NC (the four groups hold the same values); lanes 0 .. TEAM_SC - 1 a scalar each
                const bool carrier = vec && jg < NC;
                const bool scalar = lane < TEAM_SC;
                // (opaque asm between the selects: left alone, the compiler turns each chain into a per-lane indexed load from
                // a stack array -- scratch memory, or 48 KB of LDS when it promotes the array -- on the path of every exchange)
                T mine = tot[0];
#pragma unroll
                for (int i = 1; i < NC; i++) { mine = jg == i ? tot[i] : mine; asm volatile("" : "+v"(mine)); }
                double mysc = sc[0];
#pragma unroll
                for (int j = 1; j < TEAM_SC; j++) { mysc = lane == j ? sc[j] : mysc; asm volatile("" : "+v"(mysc)); }
                if (carrier) {
                    const unsigned long long b = __builtin_bit_cast(unsigned long long, (double)mine);
                    gran_store(slots + member * TEAM_GRAN + 2 * lane, (b & 0xffffffffull) | tag);
                    gran_store(slots + member * TEAM_GRAN + 2 * lane + 1, (b >> 32) | tag);
                }
                if (scalar) {
                    const unsigned long long b = __builtin_bit_cast(unsigned long long, mysc);
                    gran_store(slots + member * TEAM_GRAN + 128 + 2 * lane, (b & 0xffffffffull) | tag);
                    gran_store(slots + member * TEAM_GRAN + 129 + 2 * lane, (b >> 32) | tag);
                }
                T sum = (T)0;
                double lt = 0.0;
#pragma unroll
                for (int m = 0; m < M; m++) {
                    T pv = mine;
                    double pl = mysc;
                    if (m != member) {
#ifdef PMF_PROBE
                        const unsigned long long t_wait0 = __builtin_amdgcn_s_memtime();
#endif
                        const unsigned long long* theirs = slots + m * TEAM_GRAN;
                        unsigned long long a0 = tag, a1 = tag, l0 = tag, l1 = tag;
                        unsigned spins = 0;
                        for (;;) {
                            if (carrier) { a0 = gran_load(theirs + 2 * lane); a1 = gran_load(theirs + 2 * lane + 1); }
                            if (scalar) { l0 = gran_load(theirs + 128 + 2 * lane); l1 = gran_load(theirs + 129 + 2 * lane); }
                            const bool ok = (a0 >> 32) == xseq && (a1 >> 32) == xseq && (l0 >> 32) == xseq && (l1 >> 32) == xseq;
                            if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                            __builtin_amdgcn_s_sleep(1);
                            if ((++spins & 255u) == 0 && (spins > team_spin || __hip_atomic_load(team_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                                if (lane == 0) __hip_atomic_store(team_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                        pv = (T)__builtin_bit_cast(double, (a0 & 0xffffffffull) | (a1 << 32));
                        pl = __builtin_bit_cast(double, (l0 & 0xffffffffull) | (l1 << 32));
#ifdef PMF_PROBE
                        probe_wait += __builtin_amdgcn_s_memtime() - t_wait0;
#endif
                    }
                    sum = m == 0 ? pv : sum + pv;
                    lt = m == 0 ? pl : lt + pl;
                }
                if (carrier) team_tot[elem[jg < NC ? jg : 0]] = sum;
                if (scalar) team_l[lane] = lt;
            }
            __syncthreads();
            if (vec) {
#pragma unroll
                for (int n = 0; n < NS; n++) {
                    const SA v = *(const SA*)(team_tot + (g + G * n) * SN);
#pragma unroll
                    for (int e = 0; e < SN; e++) tot[n * SN + e] = act[n * SN + e] ? v.v[e] : (T)0;
                }
            }
#pragma unroll
            for (int j = 0; j < TEAM_SC; j++) sc[j] = team_l[j];
        }
    }
    // M > 1: TEAM_SC scalars (wave-uniform in every wave) summed over the member's waves, then over the team
    __device__ __forceinline__ void combine_scalars(double (&sc)[TEAM_SC])
    {
        if constexpr (M > 1) {
            double* red_l = (double*)(red_base + red_sel * RED_BYTES + NW * KP * sizeof(T));
            red_sel ^= 1;
            double mysc = sc[0];
#pragma unroll
            for (int j = 1; j < TEAM_SC; j++) { mysc = lane == j ? sc[j] : mysc; asm volatile("" : "+v"(mysc)); }
            if (lane < TEAM_SC) red_l[wid * TEAM_SC + lane] = mysc;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TEAM_SC; j++) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < NW; w++) t += red_l[w * TEAM_SC + j];   // same address in every lane: a broadcast read
                sc[j] = t;
            }
            T none[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) none[i] = (T)0;
            team_exchange(none, sc, false);
        }
    }

    // NW > 1: add up the NW waves' partial results (fixed order; every wave ends with the same bits)
    // vec == false (wave-uniform): only the scalar travels
    __device__ __forceinline__ void combine_waves(T (&tot)[NC], double& lsum, bool vec = true)
    {
        if constexpr (NW > 1) {
            // Alternating scratch sets: a wave may run ahead into the NEXT combine (other set) while a slow wave still
            // reads this one, but cannot reach the one after that (this set again) before everybody has passed the next
            // barrier -- so one barrier per evaluation is enough.
            T* red_part = (T*)(red_base + red_sel * RED_BYTES);
            double* red_l = (double*)(red_base + red_sel * RED_BYTES + NW * KP * sizeof(T));
            red_sel ^= 1;
            // whole 16-byte slots through LDS (the partial sums of elements past k are zero: tile and point are)
            SA* red_slots = (SA*)red_part;
            if (jg == 0 && (M == 1 || vec)) {
#pragma unroll
                for (int n = 0; n < NS; n++) {
                    SA v;
#pragma unroll
                    for (int e = 0; e < SN; e++) v.v[e] = tot[n * SN + e];
                    red_slots[wid * (G * NS) + g + G * n] = v;
                }
            }
            if (lane == 0) red_l[wid] = lsum;
            __syncthreads();
            PMF_STAMP(*this, 7);
            if constexpr (NW <= 8) {
                double lp[NW];
#pragma unroll
                for (int w = 0; w < NW; w++) lp[w] = red_l[w];
                lsum = 0.0;
#pragma unroll
                for (int w = 0; w < NW; w++) lsum += lp[w];
                if (M == 1 || vec) {
                    SA part[NW][NS];
#pragma unroll
                    for (int w = 0; w < NW; w++) {        // all reads in flight, then the sums in wave order
#pragma unroll
                        for (int n = 0; n < NS; n++) part[w][n] = red_slots[w * (G * NS) + g + G * n];
                    }
#pragma unroll
                    for (int i = 0; i < NC; i++) tot[i] = (T)0;
#pragma unroll
                    for (int w = 0; w < NW; w++) {
#pragma unroll
                        for (int n = 0; n < NS; n++) {
#pragma unroll
                            for (int e = 0; e < SN; e++) tot[n * SN + e] += act[n * SN + e] ? part[w][n].v[e] : (T)0;
                        }
                    }
                }
                if constexpr (M > 1) {
                    double sc[TEAM_SC];
#pragma unroll
                    for (int j = 0; j < TEAM_SC; j++) sc[j] = 0.0;
                    sc[0] = lsum;
                    team_exchange(tot, sc, vec);
                    lsum = sc[0];
                }
            } else {
                // (PMF_REGW16) eight waves' partials in flight at a time, sums in wave order
                lsum = 0.0;
#pragma unroll
                for (int i = 0; i < NC; i++) tot[i] = (T)0;
#pragma unroll
                for (int w0 = 0; w0 < NW; w0 += 8) {
                    SA part[8][NS];
                    double lp[8];
#pragma unroll
                    for (int w = 0; w < 8; w++) {
                        lp[w] = red_l[w0 + w];
#pragma unroll
                        for (int n = 0; n < NS; n++) part[w][n] = red_slots[(w0 + w) * (G * NS) + g + G * n];
                    }
#pragma unroll
                    for (int w = 0; w < 8; w++) {
                        lsum += lp[w];
#pragma unroll
                        for (int n = 0; n < NS; n++) {
#pragma unroll
                            for (int e = 0; e < SN; e++) tot[n * SN + e] += act[n * SN + e] ? part[w][n].v[e] : (T)0;
                        }
                    }
                }
            }
        }
    }
    __device__ __forceinline__ void gather(const T* val, unsigned nnz_row)
    {
        unsigned c0;
        my_share(nnz_row, c0, nnz);
        unsigned idx[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const unsigned j = (unsigned)(64 * b + jlane);
            idx[b] = idx_n[b];
            xr[b] = j < nnz ? val[c0 + j] : (T)0;
        }
        // byte offset of this lane's slot of factor row c: 24-bit multiply-add, 32-bit result (the host only takes
        // this engine when the factor has < 2^24 rows and < 4 GiB).  Lanes whose slot does not exist read the first
        // 16 bytes of the zero row instead.
        const unsigned rowbytes = (unsigned)ldF * (unsigned)sizeof(T);
        unsigned lane_off[NS];
#pragma unroll
        for (int n = 0; n < NS; n++) lane_off[n] = slot_on[n] ? (unsigned)((g + G * n) * 16) : 0u;
        static_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // index of nonzero JG s + jg sits in lane s % G of group jg
            const unsigned c = group_bcast<G, s % G>(idx[s / G]);
#pragma unroll
            for (int n = 0; n < NS; n++) {
                const unsigned off = __umul24(slot_on[n] ? c : zero_row, rowbytes) + lane_off[n];
                const SU v = *(const SU*)((const char*)F + (size_t)off);
#pragma unroll
                for (int e = 0; e < SN; e++) t[s][n].v[e] = v.v[e];
            }
        });
    }
    __device__ __forceinline__ void set_point(const T (&x)[NC])
    {
#pragma unroll
        for (int i = 0; i < NC; i++) a[i] = act[i] ? x[i] : (T)0;
    }

    // this lane's share of F[ind_j] . a for step s
    __device__ __forceinline__ T lane_dot(const SA (&w)[NS]) const
    {
        if constexpr (SN == 4) {
            // pk_mul + pk_fma + add per step.  (a plain mul + 3 fma -- fewer SIMD cycles on paper, a packed op takes two
            // passes on CDNA4's SIMD-32, but one more instruction to issue: C4 PG(10) 12.91 -> 13.18 ms, same box.)
            typedef T V2 __attribute__((ext_vector_type(2)));
            V2 p = (V2){ w[0].v[0], w[0].v[1] } * (V2){ a[0], a[1] };
            p = __builtin_elementwise_fma((V2){ w[0].v[2], w[0].v[3] }, (V2){ a[2], a[3] }, p);
#pragma unroll
            for (int n = 1; n < NS; n++) {
                p = __builtin_elementwise_fma((V2){ w[n].v[0], w[n].v[1] }, (V2){ a[4 * n], a[4 * n + 1] }, p);
                p = __builtin_elementwise_fma((V2){ w[n].v[2], w[n].v[3] }, (V2){ a[4 * n + 2], a[4 * n + 3] }, p);
            }
            return p.x + p.y;
        } else {
            T p = fma_t(w[0].v[1], a[1], w[0].v[0] * a[0]);
#pragma unroll
            for (int n = 1; n < NS; n++) {
                p = fma_t(w[n].v[0], a[2 * n], p);
                p = fma_t(w[n].v[1], a[2 * n + 1], p);
            }
            return p;
        }
    }

    // N (1..G) lane-partials p[u] -> lane g holds the sum over its group of p[g] (lanes g >= N: unspecified)
    template <int N> __device__ __forceinline__ T transpose_sum(const T (&p)[G]) const
    {
        T q[8], r[4], s2[2];
        constexpr int N8 = N < 8 ? N : 8;
        constexpr bool F32 = std::is_same<T, float>::value;
        if constexpr (F32 && G == 16 && N == 16) fold16_banked(p, q);             // row_ror:8           lane ^ 8
        else if constexpr (G == 16) fold_level<0x128, 8, N>(cls8, p, q);
        else {
#pragma unroll
            for (int i = 0; i < 8; i++) q[i] = p[i];
        }
        if constexpr (F32 && N8 == 8) fold8_banked(q, r);                          // row_half_mirror     lane ^ 7
        else fold_level<0x141, 4, N8>(cls4, q, r);
        constexpr int N4 = N8 < 4 ? N8 : 4;
        fold_level<0x4E, 2, N4>(cls2, r, s2);                                      // quad_perm[2,3,0,1]  lane ^ 2
        constexpr int N2 = N4 < 2 ? N4 : 2;
        if constexpr (N2 == 2) return fold_pair<0xB1>(cls1, s2[0], s2[1]);        // quad_perm[1,0,3,2]  lane ^ 1
        else return fold_one<0xB1>(s2[0]);
    }

    // add up the JG groups' partial sums (every group ends with the total) and accumulate
    __device__ __forceinline__ void combine_groups(T (&part)[NC], T (&acc)[NC]) const
    {
        if constexpr (G == 8) {
#pragma unroll
            for (int i = 0; i < NC; i++) part[i] += dpp_mov<0x128>(part[i]);       // lane ^ 8
        }
        // The exchange across the four groups.  fp32 kernels are VALU-issue bound at 2-3 waves per SIMD: ds_bpermute runs on
        // the LDS pipe and its latency hides behind the other waves (v_permlane*_swap instead: C4 PG(10) A half 5.04 ->
        // 5.46 ms, B half 7.9 -> 9.3).  fp64 kernels run one wave per SIMD and wait out every round trip: there the
        // swaps win (C3 CG fp64 A half 39.5 -> 37.3 ms).  Same bits either way.  (Swaps in the multi-wave fp32 kernels only:
        // C4 PG(10) B half 7.92 -> 8.60 ms.  Grouping all selects of a butterfly level before its DPP adds, to save the
        // s_nop wait states: A half 5.10 -> 5.81 ms -- the extra live registers cost more than the 59 s_nops per pass.
        // The matrix pipe as the adder: the four groups are the contraction index of a 16x16x4 MFMA's B operand, so
        // mfma(ones, part[i]) leaves the sum in every lane -- four MFMAs instead of eight bpermutes and eight adds, and 8-13
        // fewer registers; but 32 issue cycles each: A half 5.09 -> 5.52 ms, B half 7.92 -> 8.52.
        // eval() phase by phase over the two batches of an eight-wave kernel (all dots, all butterflies, all coefficients, all
        // axpys, so that one batch's DPP chains have the other's to interleave with): B half 7.85 -> 8.04 ms.)
        if constexpr (sizeof(T) == 8) {
#pragma unroll
            for (int i = 0; i < NC; i++) part[i] = xor_sum<16>(part[i]);
#pragma unroll
            for (int i = 0; i < NC; i++) part[i] = xor_sum<32>(part[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NC; i++) part[i] += __shfl_xor(part[i], 16);
#pragma unroll
            for (int i = 0; i < NC; i++) part[i] += __shfl_xor(part[i], 32);
        }
#pragma unroll
        for (int i = 0; i < NC; i++) acc[i] += part[i];
    }

    // Same contract as RowEval::eval (store is not supported here: pq_cap == 0)
    // FROM_CACHE: the predictions are those kept in pv (p = T.x, advanced by the accepted step): no dots, no butterfly --
    // the backward half of a pass only
    template <bool WANT_F, bool WANT_G, bool FROM_CACHE = false> __device__ __forceinline__ double eval(T sgn, T (&acc)[NC], T* store = nullptr)
    {
        n_eval++;
        PMF_STAMP(*this, 0);
#ifdef PMF_PROBE
        const unsigned long long t_eval0 = __builtin_amdgcn_s_memtime();
        struct EvalTimer { unsigned long long& acc; unsigned long long t0; __device__ ~EvalTimer() { acc += __builtin_amdgcn_s_memtime() - t0; } } eval_timer{ probe_acc[0], t_eval0 };
#endif
        double lpart = 0.0;
        T part[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) part[i] = (T)0;
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            constexpr int n = (S - G * b) < G ? (S - G * b) : G;
            T pred;
            if constexpr (FROM_CACHE && CACHED) pred = pv[b];
            else {
                T p[G];
#pragma unroll
                for (int u = 0; u < G; u++) p[u] = u < n ? lane_dot(t[(G * b + u) < S ? (G * b + u) : 0]) : (T)0;
                if constexpr (b == 0) PMF_STAMP(*this, 1);
                pred = transpose_sum<n>(p);
                if constexpr (b == 0) PMF_STAMP(*this, 2);
            }
            if constexpr (CACHED) {
                if (store == pbuf) pv[b] = pred;
                else if (store == qbuf) qv[b] = pred;
            }
            const bool on = (unsigned)(64 * b + jlane) < nnz;
            const T xj = xr[b];
            if constexpr (WANT_F) lpart += on ? (double)xj * d_log((double)pred) : 0.0;
            if constexpr (WANT_G) {
                const T coef = on ? coef_div(sgn * xj, pred) : (T)0;
                if constexpr (b == 0) PMF_STAMP(*this, 3);
                static_for<0, n>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    const T c = group_bcast<G, u>(coef);
#pragma unroll
                    for (int m = 0; m < NS; m++) {
#pragma unroll
                        for (int e = 0; e < SN; e++) part[m * SN + e] = fma_t(c, t[G * b + u][m].v[e], part[m * SN + e]);
                    }
                });
                if constexpr (b == 0) PMF_STAMP(*this, 4);
            }
        });
        PMF_STAMP(*this, 5);
        if constexpr (NW > 1 && !WANT_F && !WANT_G) {
            return 0.0;   // (the cached line search's q = T.d pass: predictions only, nothing to add up)
        } else if constexpr (NW > 1) {
            T tot[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) tot[i] = (T)0;
            if constexpr (WANT_G) combine_groups(part, tot);
            PMF_STAMP(*this, 6);
            double lsum = 0.0;
            if constexpr (WANT_F) lsum = wave_sum(lpart);
#ifdef PMF_PROBE
            const unsigned long long t_cw0 = __builtin_amdgcn_s_memtime();
#endif
            combine_waves(tot, lsum, WANT_G);
#ifdef PMF_PROBE
            probe_acc[1] += __builtin_amdgcn_s_memtime() - t_cw0;
#endif
            PMF_STAMP(*this, 8);
            if constexpr (WANT_G) {
#pragma unroll
                for (int i = 0; i < NC; i++) acc[i] += tot[i];
            }
            return lsum;
        } else {
            if constexpr (WANT_G) combine_groups(part, acc);
            PMF_STAMP(*this, 6);
            if constexpr (WANT_F) return wave_sum(lpart);
            else return 0.0;
        }
    }

    // sum_j x_j log(p_j + alpha q_j) from the cached predictions (ref: the authors' TODO at src/poismf.c:191-193)
    // `trusted` comes back false when some p_j + alpha q_j cancels to (almost) nothing: the point then sits on the boundary
    // in every coordinate that prediction depends on (k = 1: always at alpha = max_step), where the snap-to-zero of the
    // trial point -- which the cached form does not see -- decides between log(0) and log(rounding residue); the caller
    // evaluates that trial directly instead.
    __device__ __forceinline__ double logsum_cached(T alpha, bool& trusted)
    {
        double lpart = 0.0;
        bool bad = false;
#pragma unroll
        for (int b = 0; b < (CACHED ? NB : 0); b++) {
            const bool on = (unsigned)(64 * b + jlane) < nnz;
            const T pred = fma_t(alpha, qv[b], pv[b]);
            bad = bad || (on && !(pred > pv[b] * (T)1e-4));
            lpart += on ? (double)xr[b] * d_log((double)pred) : 0.0;
        }
        trusted = __builtin_amdgcn_ballot_w64(bad) == 0;
        double l = wave_sum(lpart);
        if constexpr (NW > 1) {
            // every wave of every member must take the same branch: an untrusted share poisons the sum
            if (!trusted) l = __builtin_nan("");
            T none[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) none[i] = (T)0;
            combine_waves(none, l, false);
            trusted = !(l != l);
        }
        return l;
    }
    // The same for LS_BATCH trial steps alpha, alpha decr, alpha decr^2, .. at once (teams: one exchange between CUs for what
    // would be LS_BATCH of them; an Armijo search mostly ends within the first batch)
#ifndef PMF_TEAM_LSB
#define PMF_TEAM_LSB 4
#endif
    static constexpr int LS_BATCH = M_ > 1 ? PMF_TEAM_LSB : 1;
    __device__ __forceinline__ void logsum_cached_batch(T alpha, T decr, double (&ls)[LS_BATCH], bool (&trusted)[LS_BATCH])
    {
        if constexpr (M > 1) {
#ifdef PMF_PROBE
            const unsigned long long t_ls0 = __builtin_amdgcn_s_memtime();
            struct LsTimer { unsigned long long& acc; unsigned long long t0; __device__ ~LsTimer() { acc += __builtin_amdgcn_s_memtime() - t0; } } ls_timer{ probe_acc[2], t_ls0 };
#endif
            T al = alpha;
#pragma unroll
            for (int j = 0; j < LS_BATCH; j++) {
                double lpart = 0.0;
                bool bad = false;
#pragma unroll
                for (int b = 0; b < (CACHED ? NB : 0); b++) {
                    const bool on = (unsigned)(64 * b + jlane) < nnz;
                    const T pred = fma_t(al, qv[b], pv[b]);
                    bad = bad || (on && !(pred > pv[b] * (T)1e-4));
                    lpart += on ? (double)xr[b] * d_log((double)pred) : 0.0;
                }
                const double l = wave_sum(lpart);
                ls[j] = __builtin_amdgcn_ballot_w64(bad) == 0 ? l : __builtin_nan("");
                al *= decr;
            }
            double sc[TEAM_SC];
#pragma unroll
            for (int j = 0; j < TEAM_SC; j++) sc[j] = j < LS_BATCH ? ls[j < LS_BATCH ? j : 0] : 0.0;
#ifdef PMF_PROBE
            const unsigned long long t_cs0 = __builtin_amdgcn_s_memtime();
#endif
            combine_scalars(sc);
#ifdef PMF_PROBE
            probe_acc[7] += __builtin_amdgcn_s_memtime() - t_cs0;
#endif
#pragma unroll
            for (int j = 0; j < LS_BATCH; j++) { ls[j] = sc[j]; trusted[j] = !(ls[j] != ls[j]); }
        } else {
            ls[0] = logsum_cached(alpha, trusted[0]);
        }
    }
    __device__ __forceinline__ void advance_cached(T alpha)
    {
#pragma unroll
        for (int b = 0; b < (CACHED ? NB : 0); b++) pv[b] = fma_t(alpha, qv[b], pv[b]);
    }

#ifdef PMF_PROBE
    // a value that depends on every register of the tile (the probe's "the gather has landed")
    __device__ __forceinline__ void tile_touch(T (&acc)[NC])
    {
#pragma unroll
        for (int s = 0; s < S; s++) acc[0] += t[s][0].v[0] + t[s][NS - 1].v[SN - 1];
    }
#endif
    // acc_c += sum_j F[ind_j, c]  (adjustment_Bsum's gather pass, ref: src/poismf.c:108-110)
    __device__ __forceinline__ void tile_colsum(T (&acc)[NC])
    {
        T part[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) part[i] = (T)0;
#pragma unroll
        for (int s = 0; s < S; s++) {
#pragma unroll
            for (int m = 0; m < NS; m++) {
#pragma unroll
                for (int e = 0; e < SN; e++) part[m * SN + e] += t[s][m].v[e];   // steps past the row's end hold zeros
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
        } else {
            combine_groups(part, acc);
        }
    }
};

}  // namespace pmf
