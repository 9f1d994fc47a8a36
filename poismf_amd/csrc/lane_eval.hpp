// lane_eval.hpp -- the per-row evaluation engine for DOUBLES: one nonzero per LANE, its whole factor row in that lane's registers.
//
// Same contract as RegEval / RowEval: the factor rows F[ind_j] named by a row's nonzeros are fetched once and every inner
// pass of the solver runs on chip.  The layout is the transpose of reg_eval.hpp's:
//
//   tile     t[s][c], s < L "lane sets", c < KP = 2 KS: lane l of set s holds ALL KP elements of F[ind_j], j = 64 s + l
//            (k = 50: 100 registers per set).  Nothing is padded to a power of two: 25 slots cost 25 slots (the slot
//            layout of reg_eval.hpp pays for 32), and the tile sits in architectural registers, never in AGPRs.
//   k-vector lane <-> dimension, ONE copy per wave (NC = 1 element per lane for k <= 52, 2 for k <= 104): the solver's whole
//            state is 2 NC registers per vector instead of 8, every element-wise statement of the solvers is one
//            instruction, and nothing is computed four times over (the slot layout keeps four copies per wave).
//            Which lane holds which dimension is chosen for the reduction below: with DB = KP / NC dimensions per block and
//            CW = ceil(DB / 4) columns, dimension d' of a block sits in lane (d' % CW) + 16 (d' / CW) -- k = 50: lanes 0-12
//            hold dimensions 0-12, lanes 16-28 13-25, lanes 32-44 26-38, lanes 48-58 39-49.
//   dots     pred_j = F[ind_j] . a: a plain chain of KP fused multiply-adds per lane, in the reference's own summation
//            order (ddot, left to right); `a` is read as a broadcast from LDS, one ds_read_b128 per two elements for all
//            lane sets.  No cross-lane traffic at all: the dot of nonzero j is finished in the lane that owns j.
//   axpy     g_c = sum_j coef_j F[ind_j, c]: every lane forms coef_j t[s][c] for its own nonzeros (KP multiply-adds per
//            set), and the KP lane-partials are summed over the 64 lanes by a TRANSPOSING reduction that leaves the total
//            of a dimension in the lane that holds it: per column, the four dimensions col + CW r meet in two levels of
//            v_permlane32_swap / v_permlane16_swap folds (three instructions per fold on doubles; the lane map above makes
//            every fold a full pair: 37 folds for 50 dimensions), then a CW x 16 transpose-and-add per 16-lane row through
//            LDS (CW ds_write_b64, 8 conflict-free ds_read_b128, 15 adds).
//   gather   the tile is requested with LDS-DMA (global_load_lds_dwordx4: 64 consecutive 16-byte slots of the row-major
//            image [nonzero][13 slots] per instruction, every lane's source address its own), so the fetch is as
//            coalesced as reg_eval.hpp's (13 slots = 208 contiguous bytes per nonzero) and costs no registers; each lane
//            then pulls its own nonzero's slots out of the image with ds_read_b128 (row stride 208 B: conflict-free).
//
// Instruction count of one gradient pass over a 128-nonzero tile, k = 50: 200 fp64 FMAs (dots + axpy), 144 swap-fold
// instructions, 15 adds, ~60 for the two divisions, ~50 LDS operations -- ~470 against ~950 in the slot layout
// (34 instructions per four-nonzero step, 16 of them v_accvgpr_read).
//
// This is the reference's per-nonzero ddot + daxpy (ref: src/poismf.c:126-133 calc_grad_pgd, :194-208
// calc_fun_single, :210-240 calc_grad_single[_w], :242-273 calc_fun_and_grad).
#pragma once
#include <type_traits>

#include "reg_eval.hpp"

namespace pmf {

#ifndef PMF_LANE_PREFETCH
#define PMF_LANE_PREFETCH 1
#endif

// a + b where lanes with bit 5 (W = 32) / bit 4 (W = 16) clear collect `a` and the others collect `b`: every lane ends
// with its own copy of the operand it collects plus the partner lane's (lane ^ W) copy of the same operand.
template <int W> __device__ __forceinline__ double swap_fold(double a, double b)
{
    const unsigned long long ab = __builtin_bit_cast(unsigned long long, a), bb = __builtin_bit_cast(unsigned long long, b);
    unsigned lo0, lo1, hi0, hi1;
    if constexpr (W == 32) {
        const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)ab, (unsigned)bb, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
        lo0 = rl[0]; lo1 = rl[1]; hi0 = rh[0]; hi1 = rh[1];
    } else {
        const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)ab, (unsigned)bb, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
        lo0 = rl[0]; lo1 = rl[1]; hi0 = rh[0]; hi1 = rh[1];
    }
    return __builtin_bit_cast(double, ((unsigned long long)hi0 << 32) | lo0) + __builtin_bit_cast(double, ((unsigned long long)hi1 << 32) | lo1);
}

template <class T, int KS, int L_, int NW_ = 1, int NBUF_ = 2> struct LaneEval {
    static_assert(sizeof(T) == 8, "the lane-per-nonzero engine is instantiated for doubles");
    using SA = typename Slot<T>::A;
    static constexpr int SN = Slot<T>::N;                 // elements per 16-byte slot
    static constexpr int KP = KS * SN;                    // elements of a factor row, padded to whole slots
    static constexpr int NC = (KP + WAVE - 1) / WAVE;     // elements of a k-vector per lane (dimension = lane + 64 i)
    static constexpr int L = L_, NW = NW_, M = 1, NBUF = NBUF_;
    static constexpr int W = KS < 13 ? KS : 13;           // slots per staged chunk (row stride of the LDS image: 13 slots = 52 banks, odd multiple of 4)
    static constexpr int NCH = (KS + W - 1) / W;          // chunks per factor row; chunk c starts at slot min(c W, KS - W)
    static constexpr int STAGE_BYTES = WAVE * W * 16;
    static_assert(KP % NC == 0, "blocks of equal size");
    static constexpr int DB = KP / NC;                    // dimensions per 64-lane block
    static constexpr int CW = (DB + 3) / 4;               // columns: dimension d' of a block lives in lane (d' % CW) + 16 (d' / CW)
    static_assert(CW <= 16 && DB > 2 * CW, "at most 64 dimensions per block, three or four per column");
    static constexpr int RED_STRIDE = 18 * 8;             // bytes between the 16-double rows of the transpose scratch (conflict-free b128 reads)
    static constexpr int RED_BYTES = 4 * 16 * RED_STRIDE; // four 16-lane rows x up to 16 columns
    static constexpr int AVEC_BYTES = (KP * 8 + 15) / 16 * 16;
    // NBUF == 1: the transpose scratch shares the staging buffer (eight waves per CU: 20 KB of LDS each)
    static constexpr bool ALIAS = NBUF_ == 1;
    static constexpr int WAVE_BYTES = NBUF * STAGE_BYTES + (ALIAS ? 0 : RED_BYTES) + AVEC_BYTES;
    // cross-wave scratch (NW > 1): two alternating sets of { NW x 64 NC doubles, NW scalars }
    static constexpr int XW_BYTES = NW_ > 1 ? NW_ * WAVE * NC * 8 + 16 * ((NW_ * 8 + 15) / 16) : 0;
    static constexpr int SMEM_BYTES = NW * WAVE_BYTES + 2 * XW_BYTES + 16;
    static constexpr bool PIPELINED = true;
    static constexpr bool PARKS = false;
    static constexpr bool CACHED = true, MAY_CACHE = true, CACHED_GRAD = true;
    static constexpr int LS_BATCH = 1;
    static_assert(KS >= 13 && KP <= 2 * WAVE, "25 or 50 slots");

    T t[L][KP];          // the tile
    T xr[L];             // x_j of this lane's nonzeros
    unsigned idx_n[L];   // column indices of the row whose tile is requested next (fetch_meta -> gather)
    T pv[L], qv[L];      // cached predictions p_j = F_j . x and q_j = F_j . d (solvers.hpp, cg_row_cached)
    const T* F;
    unsigned zero_row;
    int k, ldF;
    int lane, wid;
    static constexpr int member = 0;
    struct ElemOf {   // factor dimension held in element i of this lane (only meaningful where act[i])
        int d0;
        __device__ __forceinline__ int operator[](int i) const { return d0 + DB * i; }
    } elem;
    bool act[NC];
    unsigned nnz;        // nonzeros of the row held by THIS wave
    unsigned n_eval;
    unsigned char* stage;   // this wave's NBUF staging buffers
    unsigned char* red;     // this wave's transpose scratch
    SA* avec;               // this wave's copy of the current point, as slots
    unsigned char* xw_base; // NW > 1: cross-wave scratch
    int xw_sel;
    unsigned* ticket_word;
    int pq_cap;
    T* pbuf;
    T* qbuf;
#ifdef PMF_PROBE
    unsigned* probe = nullptr;
#endif

    __device__ __forceinline__ void init(const TileGeom& geo, const T* F_, unsigned char* smem)
    {
        lane = lane_id();
        wid = NW > 1 ? (int)(threadIdx.x / WAVE) : 0;
        const int col = lane & 15, rr = lane >> 4;
        elem.d0 = col + CW * rr;
        const bool lane_on = col < CW && elem.d0 < DB;
        F = F_;
        k = geo.k; ldF = geo.ldF; zero_row = geo.zero_row;
        unsigned char* p = smem + (size_t)wid * WAVE_BYTES;
        stage = p;
        red = ALIAS ? p : p + NBUF * STAGE_BYTES;
        avec = (SA*)(p + NBUF * STAGE_BYTES + (ALIAS ? 0 : RED_BYTES));
        xw_base = smem + (size_t)NW * WAVE_BYTES;
        xw_sel = 0;
        ticket_word = (unsigned*)(smem + (size_t)NW * WAVE_BYTES + 2 * XW_BYTES);
#pragma unroll
        for (int i = 0; i < NC; i++) act[i] = lane_on && elem[i] < k;
        pq_cap = 0x7fffffff;
        pbuf = (T*)(size_t)16; qbuf = (T*)(size_t)32;   // tags, never dereferenced
        n_eval = 0;
        nnz = 0;
    }
    __device__ __forceinline__ unsigned* ticket_slot() const { return ticket_word; }

    // ---- k-vectors: lane <-> dimension -------------------------------------------------------------------------------
    template <class V> __device__ __forceinline__ V rsum(V x) const { return wave_sum(x); }
    template <class V> __device__ __forceinline__ V rmin(V x) const { return wave_min(x); }
    template <class V> __device__ __forceinline__ V rmax(V x) const { return wave_max(x); }
    __device__ __forceinline__ T dot(const T (&u)[NC], const T (&v)[NC]) const
    {
        T s = (T)0;
#pragma unroll
        for (int i = 0; i < NC; i++) s = act[i] ? fma_t(u[i], v[i], s) : s;
        return rsum(s);
    }
    __device__ __forceinline__ T nrm2(const T (&u)[NC]) const { return (T)d_sqrt((double)dot(u, u)); }
    __device__ __forceinline__ void load_vec(const T* p, T (&x)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; i++) x[i] = act[i] ? p[elem[i]] : (T)0;
    }
    __device__ __forceinline__ void store_vec(T* p, const T (&x)[NC]) const
    {
        if (wid == 0) {
#pragma unroll
            for (int i = 0; i < NC; i++)
                if (act[i]) p[elem[i]] = x[i];
        }
    }
    __device__ __forceinline__ void park(int, const T (&)[NC]) {}
    __device__ __forceinline__ void unpark(int, T (&)[NC]) {}

    // ---- gather --------------------------------------------------------------------------------------------------------
    // this wave's share [c0, c0 + mine) of a row of nnz_row nonzeros (a function of the row's length and NW alone)
    __device__ __forceinline__ void my_share(unsigned nnz_row, unsigned& c0, unsigned& mine) const
    {
        if constexpr (NW > 1) {
            const unsigned C = (nnz_row + NW - 1) / NW;
            c0 = (unsigned)wid * C;
            mine = c0 < nnz_row ? (nnz_row - c0 < C ? nnz_row - c0 : C) : 0u;
        } else {
            c0 = 0u; mine = nnz_row;
        }
    }
    __device__ __forceinline__ void fetch_meta(const unsigned* ind, unsigned nnz_row)
    {
        unsigned c0, mine;
        my_share(nnz_row, c0, mine);
#pragma unroll
        for (int s = 0; s < L; s++) {
            const unsigned j = (unsigned)(WAVE * s + lane);
            idx_n[s] = j < mine ? ind[c0 + j] : zero_row;   // lanes past the end of the row fetch the all-zero row behind F
        }
    }
    __device__ __forceinline__ void begin_row(const unsigned* ind, const T* val, unsigned nnz_row)
    {
        fetch_meta(ind, nnz_row);
        gather(val, nnz_row);
    }

    // One chunk (slots [q0, q0 + W) of the 64 factor rows named by idx) -> staging buffer `buf`, as W LDS-DMA instructions.
    // Slot sigma = 64 i + lane of the row-major image [row][W slots] is row sigma / W, slot sigma % W.
    template <int Q0> __device__ __forceinline__ void dma_chunk(unsigned idx, int buf)
    {
        const unsigned rowbytes = (unsigned)ldF * (unsigned)sizeof(T);
        unsigned j4 = (unsigned)(lane / W) * 4u;            // byte address of the row's index for ds_bpermute
        unsigned q16 = (unsigned)(lane % W) * 16u;
        unsigned char* dst = stage + buf * STAGE_BYTES;
        static_for<0, W>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const unsigned c = (unsigned)__builtin_amdgcn_ds_bpermute((int)j4, (int)idx);
            const unsigned off = __umul24(c, rowbytes) + q16 + (unsigned)(Q0 * 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)F + (size_t)off),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            // next instruction: 64 slots further = 64 / W rows and 64 % W slots
            j4 += (unsigned)(WAVE / W) * 4u;
            q16 += (unsigned)(WAVE % W) * 16u;
            const bool wrap = q16 >= (unsigned)(W * 16);
            q16 = wrap ? q16 - (unsigned)(W * 16) : q16;
            j4 = wrap ? j4 + 4u : j4;
        });
    }
    // staged chunk -> this lane's slots [QLO, QHI) of set s (the chunk starts at slot Q0)
    template <int S_, int Q0, int QLO, int QHI> __device__ __forceinline__ void read_chunk(int buf)
    {
        const SA* src = (const SA*)(stage + buf * STAGE_BYTES) + lane * W;
        static_for<QLO, QHI>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const SA v = src[q - Q0];
#pragma unroll
            for (int e = 0; e < SN; e++) t[S_][q * SN + e] = v.v[e];
        });
    }
    static constexpr int chunk_start(int c) { return c * W < KS - W ? c * W : KS - W; }
    static constexpr int chunk_lo(int c) { return c == 0 ? 0 : chunk_start(c - 1) + W; }   // first slot the chunk is the first to bring
    __device__ __forceinline__ void wait_dma(int outstanding)
    {
        // LDS-DMA data is ordered for this wave's ds_reads by its own vmcnt (MI355X_MICROARCH.md, two waves per SIMD, item 7)
        if (outstanding == 0) __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0f70 | (W & 0xf) | ((W >> 4) << 14));   // vmcnt(W): the older chunk has landed
        wave_lds_fence();
    }
    __device__ __forceinline__ void gather(const T* val, unsigned nnz_row)
    {
        unsigned c0;
        my_share(nnz_row, c0, nnz);
        unsigned idx[L];
#pragma unroll
        for (int s = 0; s < L; s++) {
            const unsigned j = (unsigned)(WAVE * s + lane);
            idx[s] = idx_n[s];
            xr[s] = j < nnz ? val[c0 + j] : (T)0;
        }
        // chunks in flight: NBUF.  Work list: (set, chunk) pairs in order.
        constexpr int NWORK = L * NCH;
        static_for<0, NWORK>([&](auto wc) {
            constexpr int w = decltype(wc)::value;
            constexpr int s = w / NCH, c = w % NCH;
            if constexpr (w < NBUF) dma_chunk<chunk_start(c)>(idx[s], w % NBUF);
        });
        static_for<0, NWORK>([&](auto wc) {
            constexpr int w = decltype(wc)::value;
            constexpr int s = w / NCH, c = w % NCH;
            constexpr int inflight_after = (NWORK - 1 - w) < (NBUF - 1) ? (NWORK - 1 - w) : (NBUF - 1);
            wait_dma(inflight_after);
            read_chunk<s, chunk_start(c), chunk_lo(c), chunk_start(c) + W>(w % NBUF);
            if constexpr (w + NBUF < NWORK) {
                constexpr int w2 = w + NBUF, s2 = w2 / NCH, c2 = w2 % NCH;
                // the reads of this buffer must have returned before the next DMA lands in it
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
                wave_lds_fence();
                dma_chunk<chunk_start(c2)>(idx[s2], w % NBUF);
            }
        });
        if constexpr (ALIAS) { __builtin_amdgcn_s_waitcnt(0xc07f); wave_lds_fence(); }
    }

    __device__ __forceinline__ void set_point(const T (&x)[NC])
    {
        T* a = (T*)avec;
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < NC; i++)
            if ((lane & 15) < CW && elem.d0 < DB) a[elem[i]] = act[i] ? x[i] : (T)0;   // (dimensions k .. KP - 1 of the padded row: zeros)
        wave_lds_fence();
    }

    // ---- the transposing reduction: KP lane-partials per lane -> the total of dimension lane + 64 b in this lane -----------
    // val(c): this lane's partial of dimension c (c < KP)
    template <int C> __device__ __forceinline__ T partial(const T (&coef)[L]) const
    {
        T v = coef[0] * t[0][C];
#pragma unroll
        for (int s = 1; s < L; s++) v = fma_t(coef[s], t[s][C], v);
        return v;
    }
    // the two dimensions col + CW R0 (kept by the lanes of the lower half-wave) and col + CW (R0 + 2) (upper) of block B
    template <int B, int COL, int R0> __device__ __forceinline__ T level_a(const T (&coef)[L]) const
    {
        constexpr int d0 = COL + CW * R0, d1 = COL + CW * (R0 + 2);
        static_assert(d0 < DB, "a dimension of the block");
        const T a = partial<DB * B + d0>(coef);
        if constexpr (d1 < DB) return swap_fold<32>(a, partial<DB * B + d1>(coef));
        else return swap_fold<32>(a, a);   // (the upper half-wave's result belongs to no dimension)
    }
    template <int B> __device__ __forceinline__ T reduce_block(const T (&coef)[L])
    {
        const int R = lane >> 4, p = lane & 15;
        unsigned char* wr = red + R * (16 * RED_STRIDE) + p * 8;
        static_for<0, CW>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const T x = level_a<B, c, 0>(coef);         // rows 0 | 2
            T o;
            if constexpr (c + CW < DB) o = swap_fold<16>(x, level_a<B, c, 1>(coef));   // rows 1 | 3
            else o = swap_fold<16>(x, x);
            *(T*)(wr + c * RED_STRIDE) = o;             // lane (R, p): dimension c + CW R, summed over the four lanes (., p)
        });
        wave_lds_fence();
        const SA* rd = (const SA*)(red + R * (16 * RED_STRIDE) + p * RED_STRIDE);
        SA v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = rd[i];
        T sum = v[0].v[0];
#pragma unroll
        for (int i = 1; i < 16; i++) sum += v[i / 2].v[i % 2];
        wave_lds_fence();
        return sum;                                     // lane (R, c): dimension c + CW R (lanes with c >= CW: nothing)
    }

    // NW > 1: add up the NW waves' results (fixed order; every wave ends with the same bits)
    __device__ __forceinline__ void combine_waves(T (&tot)[NC], double& lsum, bool vec = true)
    {
        if constexpr (NW > 1) {
            T* xv = (T*)(xw_base + xw_sel * XW_BYTES);
            double* xl = (double*)(xw_base + xw_sel * XW_BYTES + NW * WAVE * NC * 8);
            xw_sel ^= 1;
            if (vec) {
#pragma unroll
                for (int i = 0; i < NC; i++) xv[(wid * NC + i) * WAVE + lane] = tot[i];
            }
            if (lane == 0) xl[wid] = lsum;
            __syncthreads();
            double lp[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) lp[w] = xl[w];
            lsum = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w++) lsum += lp[w];
            if (vec) {
                T part[NW][NC];
#pragma unroll
                for (int w = 0; w < NW; w++) {
#pragma unroll
                    for (int i = 0; i < NC; i++) part[w][i] = xv[(w * NC + i) * WAVE + lane];
                }
#pragma unroll
                for (int i = 0; i < NC; i++) {
                    T s = part[0][i];
#pragma unroll
                    for (int w = 1; w < NW; w++) s += part[w][i];
                    tot[i] = s;
                }
            }
        }
    }

    // Same contract as RegEval::eval
    template <bool WANT_F, bool WANT_G, bool FROM_CACHE = false> __device__ __forceinline__ double eval(T sgn, T (&acc)[NC], T* store = nullptr)
    {
        n_eval++;
        T pred[L];
        if constexpr (FROM_CACHE) {
#pragma unroll
            for (int s = 0; s < L; s++) pred[s] = pv[s];
        } else {
            static_for<0, KS>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const SA av = avec[q];                      // the same address in every lane: a broadcast read
#pragma unroll
                for (int s = 0; s < L; s++) {
                    if constexpr (q == 0) pred[s] = t[s][0] * av.v[0];
                    else pred[s] = fma_t(t[s][q * SN], av.v[0], pred[s]);
#pragma unroll
                    for (int e = 1; e < SN; e++) pred[s] = fma_t(t[s][q * SN + e], av.v[e], pred[s]);
                }
            });
        }
        if (store == pbuf) {
#pragma unroll
            for (int s = 0; s < L; s++) pv[s] = pred[s];
        } else if (store == qbuf) {
#pragma unroll
            for (int s = 0; s < L; s++) qv[s] = pred[s];
        }
        double lpart = 0.0;
        T coef[L];
#pragma unroll
        for (int s = 0; s < L; s++) {
            const bool on = (unsigned)(WAVE * s + lane) < nnz;
            if constexpr (WANT_F) lpart += on ? (double)xr[s] * d_log((double)pred[s]) : 0.0;
            if constexpr (WANT_G) coef[s] = on ? coef_div(sgn * xr[s], pred[s]) : (T)0;
        }
        T tot[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) tot[i] = (T)0;
        if constexpr (WANT_G) {
            static_for<0, NC>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                const T r = reduce_block<b>(coef);
                tot[b] = act[b] ? r : (T)0;
            });
        }
        if constexpr (NW > 1 && !WANT_F && !WANT_G) {
            return 0.0;
        } else if constexpr (NW > 1) {
            double lsum = 0.0;
            if constexpr (WANT_F) lsum = wave_sum(lpart);
            combine_waves(tot, lsum, WANT_G);
            if constexpr (WANT_G) {
#pragma unroll
                for (int i = 0; i < NC; i++) acc[i] += tot[i];
            }
            return lsum;
        } else {
            if constexpr (WANT_G) {
#pragma unroll
                for (int i = 0; i < NC; i++) acc[i] += tot[i];
            }
            if constexpr (WANT_F) return wave_sum(lpart);
            else return 0.0;
        }
    }

    // sum_j x_j log(p_j + alpha q_j) from the cached predictions (see RegEval::logsum_cached)
    __device__ __forceinline__ double logsum_cached(T alpha, bool& trusted)
    {
        double lpart = 0.0;
        bool bad = false;
#pragma unroll
        for (int s = 0; s < L; s++) {
            const bool on = (unsigned)(WAVE * s + lane) < nnz;
            const T pred = fma_t(alpha, qv[s], pv[s]);
            bad = bad || (on && !(pred > pv[s] * (T)1e-4));
            lpart += on ? (double)xr[s] * d_log((double)pred) : 0.0;
        }
        trusted = __builtin_amdgcn_ballot_w64(bad) == 0;
        double l = wave_sum(lpart);
        if constexpr (NW > 1) {
            if (!trusted) l = __builtin_nan("");
            T none[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) none[i] = (T)0;
            combine_waves(none, l, false);
            trusted = !(l != l);
        }
        return l;
    }
    __device__ __forceinline__ void logsum_cached_batch(T alpha, T, double (&ls)[LS_BATCH], bool (&trusted)[LS_BATCH])
    {
        ls[0] = logsum_cached(alpha, trusted[0]);
    }
    __device__ __forceinline__ void advance_cached(T alpha)
    {
#pragma unroll
        for (int s = 0; s < L; s++) pv[s] = fma_t(alpha, qv[s], pv[s]);
    }

    // acc_c += sum_j F[ind_j, c]  (adjustment_Bsum's gather pass, ref: src/poismf.c:108-110)
    __device__ __forceinline__ void tile_colsum(T (&acc)[NC])
    {
        T one[L];
#pragma unroll
        for (int s = 0; s < L; s++) one[s] = (T)1;   // lanes past the row's end hold the zero row
        T tot[NC];
        static_for<0, NC>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const T r = reduce_block<b>(one);
            tot[b] = act[b] ? r : (T)0;
        });
        if constexpr (NW > 1) {
            double unused = 0.0;
            combine_waves(tot, unused);
        }
#pragma unroll
        for (int i = 0; i < NC; i++) acc[i] += tot[i];
    }
};

}  // namespace pmf
