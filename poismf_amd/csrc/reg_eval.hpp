// reg_eval.hpp -- the per-row evaluation engine for SHORT rows: the gathered tile lives in REGISTERS.
//
// Same contract as RowEval (row_eval.hpp): one wavefront owns one output row, the factor rows F[ind_j] named by the
// row's nonzeros are fetched once and every inner pass of the solver runs on chip.  Here "on chip" is the vector
// register file instead of LDS: for rows of up to 4*S nonzeros whose factor rows fit 16 slots of 16 bytes (fp32
// k <= 64, fp64 k <= 32) the whole tile is S slots per lane (k = 50 fp32, 100 nonzeros: 100 VGPRs), there is no LDS
// traffic for the tile at all, and the waves per CU are set by registers (10-16) instead of by a 20-40 KiB LDS tile
// (4-7).  Both phases of an evaluation work on ONE layout, the slot layout the solvers keep their k-vectors in:
//
//   lane = (jg, g): jg = lane / 16 is the DPP row, g = lane % 16 the 16-byte slot of a factor row;
//   step s (0 <= s < S) handles the four nonzeros j = 4 s + jg, one per DPP row; t[s] is slot g of F[ind_j].
//
//   dots     pred_j = F[ind_j] . a : every lane multiplies its slot of t[s] with its slot of `a` (2 packed FMAs), and
//            the 16 lane-partials of a step are summed over the row.  Sixteen steps are reduced TOGETHER by a
//            transposing butterfly (4 levels: 8 + 4 + 2 + 1 select-and-add pairs instead of 16 x 4 DPP adds) that
//            leaves the finished pred of step 16 b + g in lane g -- 64 distinct nonzeros in 64 lanes, so the division
//            x_j / pred_j (and the double-precision log) is done exactly once per nonzero;
//   axpy     acc += coef_j * t[s] for s in step order: coef of step u comes from lane u of the row by ds_swizzle (the
//            LDS crossbar, no LDS memory), 2 packed FMAs per step; the four rows' partial sums are combined once per
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

// value of lane U of the caller's 16-lane row (ds_swizzle, bit-mask mode: src = (lane & 0x10) | U inside each half-wave)
template <int U> __device__ __forceinline__ int row_bcast_i32(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x10 | (U << 5)); }
template <int U> __device__ __forceinline__ unsigned row_bcast(unsigned v) { return (unsigned)row_bcast_i32<U>((int)v); }
template <int U> __device__ __forceinline__ float row_bcast(float v)
{
    return __builtin_bit_cast(float, row_bcast_i32<U>(__builtin_bit_cast(int, v)));
}
template <int U> __device__ __forceinline__ double row_bcast(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)row_bcast_i32<U>((int)(unsigned)b);
    const unsigned hi = (unsigned)row_bcast_i32<U>((int)(unsigned)(b >> 32));
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

template <class T, int S> struct RegEval {
    using SA = typename Slot<T>::A;
    using SU = typename Slot<T>::U;
    static constexpr int SN = Slot<T>::N;
    static constexpr int NC = SN;                 // one 16-byte slot per lane
    static constexpr int G = 16, JG = 4;
    static constexpr int NB = (S + 15) / 16;      // batches of 16 steps = 64 nonzeros
    static constexpr int NW = 1;

    SA t[S];        // the tile
    T a[NC];        // current point (this lane's slot)
    T xr[NB];       // x_j of the nonzero whose pred this lane finishes in batch b: j = 64 b + 4 g + jg
    // launch constants
    const T* F;
    unsigned zero_row;
    int k, s_load, tail;
    int lane, g, jg, wid;
    int jlane;      // 4 g + jg
    bool cls8, cls4, cls2, cls1;
    int elem[NC];
    bool act[NC];
    bool slot_on, slot_last;
    unsigned nnz;
    // interface parity with RowEval (the cached line search is for streamed rows only)
    int pq_cap;
    T* pbuf;
    T* qbuf;

    __device__ __forceinline__ void init(const TileGeom& geo, const T* F_, unsigned char*)
    {
        lane = lane_id();
        F = F_;
        k = geo.k; s_load = geo.s_load; zero_row = geo.zero_row;
        g = lane & 15; jg = lane >> 4; wid = 0;
        jlane = 4 * g + jg;
        cls8 = (lane & 8) != 0; cls4 = (lane & 4) != 0; cls2 = (lane & 2) != 0; cls1 = (lane & 1) != 0;
        tail = k - (s_load - 1) * SN;
        slot_on = g < s_load;
        slot_last = g == s_load - 1;
#pragma unroll
        for (int e = 0; e < SN; e++) {
            elem[e] = g * SN + e;
            act[e] = g * SN + e < k;
        }
        pq_cap = 0; pbuf = nullptr; qbuf = nullptr;
    }

    // ---- k-length vector helpers (same slot layout as RowEval with G = 16) ----------------------------
    template <class Op, class V> __device__ __forceinline__ V reduce(V x) const
    {
        x = Op::f(x, dpp_mov<0xB1>(x));
        x = Op::f(x, dpp_mov<0x4E>(x));
        x = Op::f(x, dpp_mov<0x141>(x));
        x = Op::f(x, dpp_mov<0x140>(x));
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
    __device__ __forceinline__ void load_vec(const T* p, T (&x)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; i++) x[i] = act[i] ? p[elem[i]] : (T)0;
    }
    __device__ __forceinline__ void store_vec(T* p, const T (&x)[NC]) const
    {
        if (jg == 0) {
#pragma unroll
            for (int i = 0; i < NC; i++)
                if (act[i]) p[elem[i]] = x[i];
        }
    }

    // ---- gather: indices / values coalesced in the "finishing lane" layout, then one 16-byte load per step ----
    // All S loads of a row are issued back to back, unconditionally: steps past the end of the row and lanes whose
    // slot does not exist fetch from row `zero_row` = dimF, an all-zero row the session keeps behind the factor, so
    // nothing needs masking afterwards except the excess of the last slot.
    __device__ __forceinline__ void begin_row(const unsigned* ind, const T* val, unsigned nnz_)
    {
        nnz = nnz_;
        unsigned idx[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const unsigned j = (unsigned)(64 * b + jlane);
            const bool ok = j < nnz;
            idx[b] = ok ? ind[j] : zero_row;   // steps past the end of the row fetch the all-zero row behind F
            xr[b] = ok ? val[j] : (T)0;
        }
        // byte offset of this lane's slot of factor row c: 24-bit multiply-add, 32-bit result (the host only takes
        // this engine when the factor has < 2^24 rows and < 4 GiB, see reg_engine_fits).  Lanes whose slot does not
        // exist read the first 16 bytes of the zero row instead.
        const unsigned rowbytes = (unsigned)k * (unsigned)sizeof(T);
        const unsigned lane_off = slot_on ? (unsigned)(g * 16) : 0u;
        const bool cut = slot_last && tail < SN;  // the last slot of a factor row reads past its end: zero the excess
        static_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // index of nonzero 4 s + jg sits in lane s % 16 of row jg
            const unsigned c = row_bcast<s % 16>(idx[s / 16]);
            const unsigned off = __umul24(slot_on ? c : zero_row, rowbytes) + lane_off;
            const SU v = *(const SU*)((const char*)F + (size_t)off);
            t[s].v[0] = v.v[0];
#pragma unroll
            for (int e = 1; e < SN; e++) t[s].v[e] = (cut && e >= tail) ? (T)0 : v.v[e];
        });
    }

    __device__ __forceinline__ void set_point(const T (&x)[NC])
    {
#pragma unroll
        for (int e = 0; e < SN; e++) a[e] = act[e] ? x[e] : (T)0;
    }

    // this lane's share of F[ind_j] . a for step s
    __device__ __forceinline__ T lane_dot(const SA& w) const
    {
        if constexpr (SN == 4) {
            typedef T V2 __attribute__((ext_vector_type(2)));
            V2 p = (V2){ w.v[0], w.v[1] } * (V2){ a[0], a[1] };
            p = __builtin_elementwise_fma((V2){ w.v[2], w.v[3] }, (V2){ a[2], a[3] }, p);
            return p.x + p.y;
        } else {
            return fma_t(w.v[1], a[1], w.v[0] * a[0]);
        }
    }

    // N (1..16) lane-partials p[u] -> lane g holds sum over its row of p[g] (lanes g >= N: unspecified)
    template <int N> __device__ __forceinline__ T transpose_sum(const T (&p)[16]) const
    {
        T q[8], r[4], s2[2];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (i + 8 < N) q[i] = fold_pair<0x128>(cls8, p[i], p[i + 8]);       // row_ror:8        lane ^ 8
            else if (i < N) q[i] = fold_one<0x128>(p[i]);
            else q[i] = (T)0;
        }
        constexpr int NA = N < 8 ? N : 8;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i + 4 < NA) r[i] = fold_pair<0x141>(cls4, q[i], q[i + 4]);      // row_half_mirror  lane ^ 7
            else if (i < NA) r[i] = fold_one<0x141>(q[i]);
            else r[i] = (T)0;
        }
        constexpr int NB_ = NA < 4 ? NA : 4;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (i + 2 < NB_) s2[i] = fold_pair<0x4E>(cls2, r[i], r[i + 2]);     // quad_perm[2,3,0,1]  lane ^ 2
            else if (i < NB_) s2[i] = fold_one<0x4E>(r[i]);
            else s2[i] = (T)0;
        }
        constexpr int NC_ = NB_ < 2 ? NB_ : 2;
        if (NC_ == 2) return fold_pair<0xB1>(cls1, s2[0], s2[1]);               // quad_perm[1,0,3,2]  lane ^ 1
        return fold_one<0xB1>(s2[0]);
    }

    __device__ __forceinline__ void combine_groups(T (&part)[NC], T (&acc)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; i++) part[i] += __shfl_xor(part[i], 16);
#pragma unroll
        for (int i = 0; i < NC; i++) part[i] += __shfl_xor(part[i], 32);
#pragma unroll
        for (int i = 0; i < NC; i++) acc[i] += part[i];
    }

    // Same contract as RowEval::eval (store is not supported here: pq_cap == 0)
    template <bool WANT_F, bool WANT_G> __device__ __forceinline__ double eval(T sgn, T (&acc)[NC], T* = nullptr)
    {
        double lpart = 0.0;
        T part[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) part[i] = (T)0;
        static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            constexpr int n = (S - 16 * b) < 16 ? (S - 16 * b) : 16;
            T p[16];
#pragma unroll
            for (int u = 0; u < 16; u++) p[u] = u < n ? lane_dot(t[(16 * b + u) < S ? (16 * b + u) : 0]) : (T)0;
            const T pred = transpose_sum<n>(p);
            const bool on = (unsigned)(64 * b + jlane) < nnz;
            const T xj = xr[b];
            if constexpr (WANT_F) lpart += on ? (double)xj * d_log((double)pred) : 0.0;
            if constexpr (WANT_G) {
                const T coef = on ? sgn * xj / pred : (T)0;
                static_for<0, n>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    const T c = row_bcast<u>(coef);
                    const SA& w = t[16 * b + u];
#pragma unroll
                    for (int e = 0; e < SN; e++) part[e] = fma_t(c, w.v[e], part[e]);
                });
            }
        });
        if constexpr (WANT_G) combine_groups(part, acc);
        if constexpr (WANT_F) return wave_sum(lpart);
        else return 0.0;
    }

    __device__ __forceinline__ double logsum_cached(T) const { return 0.0; }
    __device__ __forceinline__ void advance_cached(T) {}

    // acc_c += sum_j F[ind_j, c]  (adjustment_Bsum's gather pass, ref: src/poismf.c:108-110)
    __device__ __forceinline__ void tile_colsum(T (&acc)[NC])
    {
        T part[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) part[i] = (T)0;
#pragma unroll
        for (int s = 0; s < S; s++) {
#pragma unroll
            for (int e = 0; e < SN; e++) part[e] += t[s].v[e];   // slots of steps past the row's end are zero
        }
        combine_groups(part, acc);
    }
};

}  // namespace pmf
