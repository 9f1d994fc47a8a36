// row_eval.hpp -- the per-row evaluation engine shared by the three solvers.
//
// One wavefront owns one output row r of the factor being updated (M) against the fixed opposing
// factor F.  The rows F[ind_j] named by the row's nonzeros are gathered ONCE from HBM/L2 into an LDS
// tile with 16-byte global loads (4-byte aligned, ~13 lanes per 200-byte row so every load
// instruction moves up to 1 KiB) and every inner pass of the solver then runs from LDS:
//
//   phase 1  lane <-> nonzero : pred_j = T[j,:] . a     (ds_read_b128 of the lane's own tile row; the
//                                                        row stride is an ODD number of 16-byte slots,
//                                                        so the 16 lanes of a b128 group hit 16 slots)
//   phase 2  lane <-> factor dimension : acc_c += coef_j * T[j,c] for j in nonzero order (conflict-free
//                                                        ds_read, coef_j broadcast with v_readlane)
//
// which is what the reference does per nonzero with one ddot + one daxpy
// (ref: src/poismf.c:126-133 calc_grad_pgd, :194-208 calc_fun_single, :210-240 calc_grad_single[_w],
// :242-273 calc_fun_and_grad), but with no per-nonzero cross-lane reduction and no re-gather.
// Rows that do not fit the tile (cap) are streamed chunk by chunk on every pass instead.
#pragma once
#include "wave_ops.hpp"

namespace pmf {

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

// Geometry of one launch (host fills it in; see plan_launch in poismf_hip.hip).
struct TileGeom {
    int k;         // factor dimension
    int s_load;    // 16-byte slots actually holding data per factor row = ceil(k*sizeof(T)/16)
    int s_stride;  // LDS row stride in slots = s_load | 1 (odd: conflict-free ds_read_b128 down a column)
    int cap;       // nonzeros the tile can hold
    int resident;  // 1: every row of this launch has nnz <= cap, gather once per row
};

__host__ __device__ inline size_t lds_bytes_per_wave(const TileGeom& g, size_t sizeof_real)
{
    size_t b = (size_t)g.cap * g.s_stride * 16;          // tile
    b += (size_t)g.s_load * 16;                          // current point a (padded with zeros)
    b += (((size_t)g.cap * sizeof_real) + 15) / 16 * 16; // x_j
    b += (((size_t)g.cap * 4) + 15) / 16 * 16;           // ind_j
    return b;
}

template <class T, int NC> struct RowEval {
    using SA = typename Slot<T>::A;
    using SU = typename Slot<T>::U;
    static constexpr int SN = Slot<T>::N;

    // LDS carve-out of this wave
    SA* tile;
    SA* avec;
    T* xb;
    unsigned* idxb;
    // launch constants
    const T* F;
    int k, s_load, s_stride, cap, tail;
    bool resident;
    int lane;
    int gj0, gt0, gdj, gdt;  // lane -> (nonzero, slot) walk of the gather, advanced 64 slots at a time
    int coff[NC];            // element offsets lane + 64 i clamped into the row
    bool act[NC];            // lane + 64 i < k
    // current row
    const unsigned* ind;
    const T* val;
    unsigned nnz;

    __device__ __forceinline__ void init(const TileGeom& g, const T* F_, unsigned char* smem)
    {
        lane = lane_id();
        F = F_;
        k = g.k; s_load = g.s_load; s_stride = g.s_stride; cap = g.cap; resident = g.resident != 0;
        tail = k - (s_load - 1) * SN;  // valid elements in the last slot of a factor row (1..SN)
        unsigned char* p = smem;
        tile = (SA*)p; p += (size_t)cap * s_stride * 16;
        avec = (SA*)p; p += (size_t)s_load * 16;
        xb = (T*)p; p += (((size_t)cap * sizeof(T)) + 15) / 16 * 16;
        idxb = (unsigned*)p;
        gj0 = lane / s_load; gt0 = lane % s_load;
        gdj = WAVE / s_load; gdt = WAVE % s_load;
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const int c = lane + WAVE * i;
            act[i] = c < k;
            coff[i] = act[i] ? c : k - 1;
        }
        // zero the padding of the point vector once; set_point only ever writes the first k entries
        for (int c = k + lane; c < s_load * SN; c += WAVE) ((T*)avec)[c] = (T)0;
        wave_lds_fence();
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
            v[u] = *(const SU*)(F + (size_t)col * (size_t)k + (size_t)(tr * SN));
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

    // Gather chunk [c0, c0+cn) of the current row: indices and values, then the factor rows.
    __device__ __forceinline__ void load_chunk(unsigned c0, int cn)
    {
        for (int j = lane; j < cn; j += WAVE) {
            idxb[j] = ind[c0 + j];
            xb[j] = val[c0 + j];
        }
        wave_lds_fence();
        const int Q = cn * s_load;  // 16-byte slots to fetch
        int j = gj0, t = gt0;
        int q0 = 0;
        for (; q0 + 8 * WAVE <= Q; q0 += 8 * WAVE) gather_batch<8>(q0, Q, j, t);  // 8 KiB in flight per wave
        const int rem = Q - q0;
        if (rem > 4 * WAVE) gather_batch<8>(q0, Q, j, t);
        else if (rem > 2 * WAVE) gather_batch<4>(q0, Q, j, t);
        else if (rem > WAVE) gather_batch<2>(q0, Q, j, t);
        else if (rem > 0) gather_batch<1>(q0, Q, j, t);
        wave_lds_fence();
    }

    __device__ __forceinline__ void begin_row(const unsigned* ind_, const T* val_, unsigned nnz_)
    {
        ind = ind_; val = val_; nnz = nnz_;
        if (resident && nnz > 0) load_chunk(0, (int)nnz);
    }

    // Publish the point at which the next evaluations happen.
    __device__ __forceinline__ void set_point(const T (&x)[NC])
    {
#pragma unroll
        for (int i = 0; i < NC; i++)
            if (act[i]) ((T*)avec)[lane + WAVE * i] = x[i];
        wave_lds_fence();
    }

    // phase 1: this lane's nonzero (jb + lane) of the loaded chunk
    __device__ __forceinline__ T pred_lane(int jb, int cn) const
    {
        const int j = jb + lane;
        const SA* row = tile + (size_t)(j < cn ? j : cn - 1) * s_stride;
        T p[SN];
#pragma unroll
        for (int e = 0; e < SN; e++) p[e] = (T)0;
        // blocks of 4 slots with all 8 LDS reads issued before the first use, then the 0-3 left over
        int t = 0;
        for (; t + 4 <= s_load; t += 4) {
            SA tv[4], av[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { tv[u] = row[t + u]; av[u] = avec[t + u]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int e = 0; e < SN; e++) p[e] = fma_t(tv[u].v[e], av[u].v[e], p[e]);
            }
        }
        {
            const int rem = s_load - t;  // 0..3, wave-uniform
            SA tv[3], av[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int tt = (u < rem) ? t + u : t;  // clamped: a repeated slot, multiplied by zero below
                tv[u] = row[tt < s_load ? tt : 0];
                av[u] = avec[tt < s_load ? tt : 0];
            }
#pragma unroll
            for (int u = 0; u < 3; u++) {
                if (u < rem) {
#pragma unroll
                    for (int e = 0; e < SN; e++) p[e] = fma_t(tv[u].v[e], av[u].v[e], p[e]);
                }
            }
        }
        if constexpr (SN == 4) return (p[0] + p[1]) + (p[2] + p[3]);
        else return p[0] + p[1];
    }

    // phase 2: acc_c += sum over the cnt nonzeros starting at jb of coef_j * T[j, c], in nonzero order.
    // Blocks of 8 are written out by hand: the loop contains a convergent operation (v_readlane), so the
    // compiler will not unroll it on its own, and an un-unrolled body pays one full LDS latency per nonzero.
    __device__ __forceinline__ void accumulate(int jb, int cnt, T coef, T (&acc)[NC]) const
    {
        const int rs = s_stride * SN;
        const T* tf = (const T*)tile + (size_t)jb * (size_t)rs;
        int jj = 0;
        for (; jj + 8 <= cnt; jj += 8) {
            T tv[8][NC];
            T cj[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int i = 0; i < NC; i++) tv[u][i] = tf[(jj + u) * rs + coff[i]];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) cj[u] = read_lane(coef, jj + u);
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int i = 0; i < NC; i++) acc[i] = fma_t(cj[u], tv[u][i], acc[i]);
            }
        }
        for (; jj < cnt; jj++) {
            const T c1 = read_lane(coef, jj);
#pragma unroll
            for (int i = 0; i < NC; i++) acc[i] = fma_t(c1, tf[jj * rs + coff[i]], acc[i]);
        }
    }

    // At the point last published with set_point:
    //   WANT_F : returns lsum = sum_j x_j log(pred_j)   (log and the sum in double, as the reference's
    //            `lsum += X[ix] * log(dot)` is a double expression even in its float build)
    //   WANT_G : acc_c += sum_j (sgn x_j / pred_j) F[ind_j, c]  in nonzero order
    template <bool WANT_F, bool WANT_G> __device__ __forceinline__ double eval(T sgn, T (&acc)[NC])
    {
        double lpart = 0.0;
        for (unsigned c0 = 0; c0 < nnz; c0 += (unsigned)cap) {
            const int cn = (int)((nnz - c0 < (unsigned)cap) ? nnz - c0 : (unsigned)cap);
            if (!resident) load_chunk(c0, cn);
            for (int jb = 0; jb < cn; jb += WAVE) {
                const T pred = pred_lane(jb, cn);
                const bool on = jb + lane < cn;
                const T xj = xb[on ? jb + lane : 0];
                if constexpr (WANT_F) lpart += on ? (double)xj * d_log((double)pred) : 0.0;
                if constexpr (WANT_G) {
                    const T coef = on ? sgn * xj / pred : (T)0;
                    accumulate(jb, cn - jb < WAVE ? cn - jb : WAVE, coef, acc);
                }
            }
        }
        if constexpr (WANT_F) return wave_sum(lpart);
        else return 0.0;
    }

    // acc_c += sum_j F[ind_j, c]   (the gather pass of adjustment_Bsum, ref: src/poismf.c:108-110,
    // served from the tile instead of a second trip to memory)
    __device__ __forceinline__ void tile_colsum(T (&acc)[NC])
    {
        for (unsigned c0 = 0; c0 < nnz; c0 += (unsigned)cap) {
            const int cn = (int)((nnz - c0 < (unsigned)cap) ? nnz - c0 : (unsigned)cap);
            if (!resident) load_chunk(c0, cn);
            for (int jb = 0; jb < cn; jb += WAVE)
                accumulate(jb, cn - jb < WAVE ? cn - jb : WAVE, (T)1, acc);
        }
    }

    // ---- k-length vector helpers on the lane <-> dimension layout --------------------------------
    __device__ __forceinline__ T dot(const T (&u)[NC], const T (&v)[NC]) const
    {
        T s = (T)0;
#pragma unroll
        for (int i = 0; i < NC; i++) s = act[i] ? fma_t(u[i], v[i], s) : s;
        return wave_sum(s);
    }
    __device__ __forceinline__ T nrm2(const T (&u)[NC]) const { return (T)d_sqrt((double)dot(u, u)); }

    __device__ __forceinline__ void load_vec(const T* p, T (&x)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; i++) x[i] = act[i] ? p[lane + WAVE * i] : (T)0;
    }
    __device__ __forceinline__ void store_vec(T* p, const T (&x)[NC]) const
    {
#pragma unroll
        for (int i = 0; i < NC; i++)
            if (act[i]) p[lane + WAVE * i] = x[i];
    }
};

}  // namespace pmf
