// wave_ops.hpp -- 64-lane wavefront primitives for gfx950 (CDNA4).
//
// Every row sub-problem is owned by ONE wavefront, so all solver scalars are wave-uniform and every
// k-length reduction is a cross-lane reduction: 4 DPP steps inside each 16-lane row (v_add_*_dpp,
// no LDS traffic), then 4 v_readlane + 3 adds across the rows.  Results are returned through
// readfirstlane so the compiler knows they are uniform (scalar branches, SGPR operands).
#pragma once
#include <hip/hip_runtime.h>

namespace pmf {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }

// ---- raw lane movement on 32- and 64-bit payloads -------------------------------------------
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v)
{
    // bound_ctrl: lanes without a source lane read 0 -- what `old` = 0 would give them too, but with every lane written the
    // compiler needs no v_mov to set `old` up first (doubles: two of them per move, a seventh of the fp64 evaluation)
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)dpp_i32<CTRL>((int)(unsigned)b);
    const unsigned hi = (unsigned)dpp_i32<CTRL>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) { return dpp_i32<CTRL>(v); }

__device__ __forceinline__ float read_lane(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double read_lane(double v, int l)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int read_lane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

__device__ __forceinline__ float uniform(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double uniform(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ bool uniform(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }

// ---- x[lane] + x[lane ^ 16] and x[lane] + x[lane ^ 32] without the LDS crossbar ------------------
// gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd 16-lane rows (the upper half-wave) of one register
// with the even rows (the lower half) of another: with both operands holding x, the two results are "the even rows' x
// everywhere" and "the odd rows' x everywhere", whose sum is x + x[lane ^ 16] (^ 32) in every lane -- one VALU
// instruction instead of a ds_bpermute round trip (~100 cycles) per exchange.  Same bits as x + __shfl_xor(x, 16):
// the two addends are the same pair in either order.
__device__ __forceinline__ unsigned swap_sum_bits16(unsigned v, unsigned& other)
{
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    other = r[1];
    return r[0];
}
__device__ __forceinline__ unsigned swap_sum_bits32(unsigned v, unsigned& other)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    other = r[1];
    return r[0];
}
template <int W> __device__ __forceinline__ float xor_sum(float x)
{
    unsigned o;
    const unsigned e = W == 16 ? swap_sum_bits16(__builtin_bit_cast(unsigned, x), o) : swap_sum_bits32(__builtin_bit_cast(unsigned, x), o);
    return __builtin_bit_cast(float, e) + __builtin_bit_cast(float, o);
}
template <int W> __device__ __forceinline__ double xor_sum(double x)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    unsigned olo, ohi;
    const unsigned elo = W == 16 ? swap_sum_bits16((unsigned)b, olo) : swap_sum_bits32((unsigned)b, olo);
    const unsigned ehi = W == 16 ? swap_sum_bits16((unsigned)(b >> 32), ohi) : swap_sum_bits32((unsigned)(b >> 32), ohi);
    return __builtin_bit_cast(double, ((unsigned long long)ehi << 32) | elo) + __builtin_bit_cast(double, ((unsigned long long)ohi << 32) | olo);
}

// ---- reductions ---------------------------------------------------------------------------------
struct OpSum { template <class T> static __device__ __forceinline__ T f(T a, T b) { return a + b; } };
struct OpMin {
    static __device__ __forceinline__ float f(float a, float b) { return fminf(a, b); }
    static __device__ __forceinline__ double f(double a, double b) { return fmin(a, b); }
    static __device__ __forceinline__ int f(int a, int b) { return a < b ? a : b; }
};
struct OpMax {
    static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); }
    static __device__ __forceinline__ double f(double a, double b) { return fmax(a, b); }
    static __device__ __forceinline__ int f(int a, int b) { return a > b ? a : b; }
};

// DPP controls: quad_perm[1,0,3,2] = 0xB1, quad_perm[2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140.  After the four steps every lane of a 16-lane row holds that row's result.
template <class Op, class T> __device__ __forceinline__ T wave_reduce(T x)
{
    x = Op::f(x, dpp_mov<0xB1>(x));
    x = Op::f(x, dpp_mov<0x4E>(x));
    x = Op::f(x, dpp_mov<0x141>(x));
    x = Op::f(x, dpp_mov<0x140>(x));
    const T r0 = read_lane(x, 0), r1 = read_lane(x, 16), r2 = read_lane(x, 32), r3 = read_lane(x, 48);
    return uniform(Op::f(Op::f(r0, r1), Op::f(r2, r3)));
}
template <class T> __device__ __forceinline__ T wave_sum(T x) { return wave_reduce<OpSum>(x); }
template <class T> __device__ __forceinline__ T wave_min(T x) { return wave_reduce<OpMin>(x); }
template <class T> __device__ __forceinline__ T wave_max(T x) { return wave_reduce<OpMax>(x); }

// Orders this wave's LDS traffic: data written by some lanes is read by other lanes of the SAME wave.
// The hardware executes one wave's DS instructions in issue order; this keeps the compiler from
// reordering them and is free at run time.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Fused multiply-add in the working precision (__builtin_fma alone is the DOUBLE builtin: on floats it
// would convert, do an f64 fma and convert back).
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Double-typed libm wrappers.  The reference's C sources call fabs/sqrt/log/log10/fmin/fmax, which in
// C are the DOUBLE functions even in its float build (USE_FLOAT); C++ overload resolution would pick
// the float overloads and change the rounding of the surrounding expressions.
__device__ __forceinline__ double d_abs(double v) { return fabs(v); }
__device__ __forceinline__ double d_sqrt(double v) { return sqrt(v); }
// log in double for the likelihood sums x_j log(pred_j): with ~27 function evaluations per row and half-sweep in CG
// (and one per evaluation in TNCG) it is the largest single item of those solvers, and the device library's
// double-double implementation costs 98 VALU instructions.  This is the classic fdlibm scheme (Sun's e_log.c:
// x = 2^e m, m in [sqrt(1/2), sqrt 2), f = m - 1, s = f / (2 + f), log m = f - f^2/2 + s (f^2/2 + R(s^2)) with a
// degree-7 minimax R, e ln2 added in a hi/lo split; error < 1 ulp) with the division done as v_rcp_f64 + two Newton
// steps + one residual correction: 45 instructions.  Zero, negative, infinite and NaN arguments return what log()
// returns.  poismf_hip_selftest_log() compares it with the device library on the device.
__device__ __forceinline__ double d_log_lib(double v) { return log(v); }
__device__ __forceinline__ double d_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int e = __builtin_amdgcn_frexp_exp(x);          // x = m 2^e, m in [1/2, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;                            // m in [sqrt(1/2), sqrt 2)
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    double s = f * y;
    s = __builtin_fma(__builtin_fma(-d, s, f), y, s);
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    double r = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    // x = +inf -> +inf; x = 0 -> -inf; x < 0 or NaN -> NaN
    r = (x == __builtin_inf()) ? x : r;
    r = (x == 0.0) ? -__builtin_inf() : r;
    r = (x < 0.0 || x != x) ? __builtin_nan("") : r;
    return r;
}
__device__ __forceinline__ double d_log10(double v) { return log10(v); }
__device__ __forceinline__ double d_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double d_max(double a, double b) { return fmax(a, b); }

template <class T> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1.1920928955078125e-07f; };   // FLT_EPSILON
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };   // DBL_EPSILON

}  // namespace pmf
