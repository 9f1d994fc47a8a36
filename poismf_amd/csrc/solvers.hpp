// solvers.hpp -- the three per-row inner solvers as wave-resident state machines.
//
// State vectors live in registers in the slot layout of row_eval.hpp (NC elements per lane = whole 16-byte
// slots, JG identical copies per wave); every scalar is wave-uniform, so every data-dependent branch of the line searches is a scalar
// branch with no divergence.  Arithmetic follows the reference statement by statement, including the
// places where its C sources promote to double inside the float build.
//
//   pg_row   : ref src/poismf.c:172-185 (the maxupd loop of pg_iteration)
//   cg_row   : ref src/nonnegcg.c:177-346 (minimize_nonneg_cg) with cg_iteration's constants, src/poismf.c:315-320
//   tnc_row  : ref src/tnc.c:251-463 (tnc), :554-993 (tnc_minimize), :1162-1341 (tnc_direction),
//              :1388-1435 (hessianTimesVector), :1444-1528 (msolve), :1580-1658 (initPreconditioner),
//              :1664-1813 (linearSearch), :1822-2154 (getptcInit/getptcIter), with tncg_iteration's
//              constants, src/poismf.c:383-391
#pragma once
#include <type_traits>
#include <utility>

#include "row_eval.hpp"

namespace pmf {

#define PMF_EW _Pragma("unroll") for (int i = 0; i < NC; i++)

template <class T> __device__ __forceinline__ bool not_finite(T v) { return isnan(v) || isinf(v); }

// What a row's solver decided, for tests that pin the decisions and not only the result (the reference hands the same numbers
// back from minimize_nonneg_cg -- niter, nfeval, ref: src/nonnegcg.c:177-189 -- and from tnc -- nfeval, niter, rc, ref:
// src/tnc.c:251-260; cg_iteration / tncg_iteration drop them).  CG rc: 0 |g.d| <= tol, 1 evaluation budget, 2 iteration
// budget, 3 f(x0) not finite.  Wave-uniform scalars; written to memory only in profiling sessions.
struct SolveStats { int niter = 0, nfeval = 0, rc = 0; };

// ------------------------------------------------------------------------------------------------
// Proximal gradient
// ------------------------------------------------------------------------------------------------
template <class EV, class T, int NC>
__device__ __forceinline__ void pg_row(EV& ev, const RowParams<T>& P, T (&x)[NC], const T (&shift)[NC])
{
    for (int u = 0; u < P.maxupd; u++) {
        ev.set_point(x);
        T g[NC];
        PMF_EW g[i] = (T)0;
        ev.template eval<false, true>((T)1, g);                   // calc_grad_pgd, ref: :126-133
        PMF_EW {
            x[i] = fma_t(P.step, g[i], x[i]);              // a += step * grad
            x[i] = x[i] + shift[i];                                // a += (pre-scaled) Bsum
            x[i] = x[i] * P.cnst_div;                              // a *= 1 / (1 + 2 l2 step)
            x[i] = (x[i] > (T)0) ? x[i] : (T)0;                    // a = max(a, 0)
        }
        PMF_STAMP(ev, 9);
    }
}

// ------------------------------------------------------------------------------------------------
// Objective / gradient wrappers on top of RowEval
// ------------------------------------------------------------------------------------------------
// f = bsum.a + l2 (a.a) - w sum x log(a.F_j)                     ref: src/poismf.c:194-208
template <class EV, class T, int NC>
__device__ __forceinline__ T fun_single(EV& ev, const RowParams<T>& P, const T (&bsum)[NC], const T (&a)[NC])
{
    ev.set_point(a);
    T dummy[NC];
    PMF_EW dummy[i] = (T)0;
    T reg = ev.dot(bsum, a);
    reg += P.l2 * ev.dot(a, a);
    const T lsum = (T)ev.template eval<true, false>((T)0, dummy);
    return reg - lsum * P.w;
}

// w == 1: g = bsum + 2 l2 a - sum (x/(a.F_j)) F_j                 ref: src/poismf.c:210-223
// w != 1: g = w (-sum ...) + bsum_row + 2 l2 a                    ref: src/poismf.c:225-240 (quirk Q11)
template <class EV, class T, int NC>
__device__ __forceinline__ void grad_single(EV& ev, const RowParams<T>& P, const T (&bsum)[NC],
                                            const T (&a)[NC], T (&g)[NC], bool weighted)
{
    ev.set_point(a);
    const T two_l2 = (T)(2. * (double)P.l2);
    if (!weighted) {
        PMF_EW g[i] = fma_t(two_l2, a[i], bsum[i]);
        ev.template eval<false, true>((T)-1, g);
    } else {
        PMF_EW g[i] = (T)0;
        ev.template eval<false, true>((T)-1, g);
        PMF_EW {
            g[i] = g[i] * P.w;
            g[i] = g[i] + bsum[i];
            g[i] = fma_t(two_l2, a[i], g[i]);
        }
    }
}

// fused f and g for TNC; f omits the l2 term (quirk Q4)           ref: src/poismf.c:242-273
template <class EV, class T, int NC>
__device__ __forceinline__ T fun_and_grad(EV& ev, const RowParams<T>& P, const T (&bsum)[NC],
                                          const T (&a)[NC], T (&g)[NC])
{
    ev.set_point(a);
#ifdef PMF_DUP_EVAL   // development: every TNC evaluation made twice (same results and decisions): the difference to the plain build is what the evaluations cost
    {
        T g2[NC];
        PMF_EW g2[i] = (T)0;
        const double l2nd = ev.template eval<true, true>((T)-1, g2);
        asm volatile("" :: "v"(g2[0]), "v"(l2nd));
        ev.n_eval--;
    }
#endif
    PMF_EW g[i] = (T)0;
    const T lsum = (T)ev.template eval<true, true>((T)-1, g);
    const T two_l2 = (T)(2. * (double)P.l2);
    if (P.w != (T)1) { PMF_EW g[i] = g[i] * P.w; }
    PMF_EW g[i] = g[i] + bsum[i];
    const T reg = ev.dot(bsum, a);
    PMF_EW g[i] = fma_t(two_l2, a[i], g[i]);
    return reg - lsum * P.w;
}

// ------------------------------------------------------------------------------------------------
// Line-search trials that are certain to fail, without evaluating them.
// The row objective is f(a) = r(a) - w sum_j x_j log(a . F_j) with r quadratic (Bsum . a + l2 a . a) and -- for x_j > 0, i.e.
// counts -- a data term that is CONCAVE along any line; with `limit_step` the trial point is x + s d itself (feasible; only
// components below 1e-15 are snapped to 0).  The tangent of the concave part at s = 0 therefore bounds f from below:
//     f(x + s d) >= f(x) + s g.d + s^2 l2 d.d
// A trial whose lower bound already misses the Armijo threshold f_cur - c s d.d by more than any rounding error fails in the
// reference too (ref: src/nonnegcg.c:310-320: nfeval++, step *= decr): it is counted and skipped, no logarithm taken.  With
// cg_iteration's l2 = 1e4 (poismf/__init__.py:250) that is every step above ~1e-4 -- the first five to seven of the ~six trials
// an iteration takes from max_step <= 1 by factors of 4.  The last of the max_ls trials is always evaluated (quirk Q2 hands its
// value on), and f_x is f at the current x even when f_cur is not (Q2 again).
// Returns false when the evaluation budget runs out on the way (the reference returns from the row there, ref: :316-320).
// Assumptions: x_j > 0 AND w > 0 (RowParams::x_pos: the host clears it for w <= 0, where the data term is not convex along the
// line); and the trial point is x + s d up to the snap of components below 1e-15 to 0 (quirk Q10).  The snap can only RAISE the
// data term (a prediction gets smaller, -w x_j log(.) larger) and moves the quadratic part by at most |Bsum_i + 2 l2 x_i| 1e-15 per
// snapped coordinate, which the 1e-9 / 1e-3 relative margin covers: the bound stays a lower bound of the value the reference sees.
template <class T>
__device__ __forceinline__ bool skip_certain_failures(T f_x, T f_cur, T gd, T l2dd, T dd, T c_ls, T decr, int max_ls, int maxnfeval,
                                                      T& step, int& ls, int& nfeval)
{
    const T eps_m = sizeof(T) == 8 ? (T)1e-9 : (T)1e-3;   // >> the rounding error of any of these objective values
    const T marg = eps_m * (T)(d_abs((double)f_x) + d_abs((double)f_cur));
    while (ls < max_ls - 1) {
        const T lb = fma_t(step * step, l2dd, fma_t(step, gd, f_x));
        const T thr = f_cur - c_ls * step * dd;
        if (!(lb > thr + marg)) break;
        nfeval++;
        if (nfeval >= maxnfeval) return false;
        step *= decr;
        ls++;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Non-negative Polak-Ribiere CG (Li 2013)                        ref: src/nonnegcg.c:177-346
// ------------------------------------------------------------------------------------------------
template <class EV, class T, int NC>
__device__ __forceinline__ void cg_row(EV& ev, const RowParams<T>& P, const T (&bsum)[NC], T (&x)[NC],
                                       bool weighted, SolveStats& st)
{
    const T tol = (T)1e-2, decr = (T)0.25, c_ls = (T)0.01;         // ref: src/poismf.c:318-319
    const int maxnfeval = 150, max_ls = 20;
    const int maxiter = P.maxupd > 0 ? P.maxupd : 0x7fffffff;
    T g[NC], d[NC], gp[NC], dp[NC], trial[NC];
    PMF_EW { gp[i] = (T)0; dp[i] = (T)0; }
    T gprev_sq = (T)0;
    T f_cur = fun_single(ev, P, bsum, x);                          // ref: :191
    T f_new = (T)0;
    T f_x = f_cur;                                                 // f at the current x (f_cur may be a refused point's value, quirk Q2)
    int nfeval = 1;
    int it = 0;
    auto leave = [&](int rc) { st.niter = it; st.nfeval = nfeval; st.rc = rc; };
    if (not_finite(f_cur)) { leave(3); return; }                   // ref: :223-226, row left unchanged
    const bool prune = P.x_pos != 0 && P.limit_step != 0;

    for (it = 0; it < maxiter; it++) {
        grad_single(ev, P, bsum, x, g, weighted);                  // ref: :231
        PMF_EW d[i] = (x[i] <= (T)0 && g[i] >= (T)0) ? (T)0 : -g[i];   // ref: :236-239
        if (it > 0) {                                              // ref: :242-261
            T th = (T)0, be = (T)0;
            PMF_EW {
                const bool on = ev.act[i] && !(x[i] <= (T)0);
                th += on ? g[i] * dp[i] : (T)0;
                be += on ? g[i] * (g[i] - gp[i]) : (T)0;
            }
            T theta = ev.rsum(th), beta = ev.rsum(be);
            theta /= gprev_sq;
            beta /= gprev_sq;
            PMF_EW d[i] += (x[i] <= (T)0) ? (T)0 : beta * dp[i] - theta * (g[i] - gp[i]);
        }
        const T gd = ev.dot(g, d);                                 // ref: :264-269
        if (d_abs((double)gd) <= (double)tol) { leave(0); return; }

        T max_step;                                                // ref: :272-288
        if (P.limit_step) {
            T m = (T)1;
            PMF_EW if (ev.act[i] && d[i] < (T)0) m = (T)d_min((double)m, (double)(-x[i] / d[i]));
            max_step = ev.rmin(m);
        } else {
            T m = (T)0;
            PMF_EW if (ev.act[i] && d[i] < (T)0) m = (T)d_max((double)m, (double)(-x[i] / d[i]));
            max_step = ev.rmax(m);
            max_step = (T)d_min(1., 0.99 * (double)max_step);
        }

        const T dd = ev.dot(d, d);                                 // ref: :295
        T step = max_step;
        int ls = 0;
        if (prune && !skip_certain_failures(f_x, f_cur, gd, P.l2 * dd, dd, c_ls, decr, max_ls, maxnfeval, step, ls, nfeval)) { leave(1); return; }
        for (; ls < max_ls; ls++) {                                // ref: :297-327
            PMF_EW {
                trial[i] = fma_t(step, d[i], x[i]);
                if (P.limit_step) trial[i] = ((double)trial[i] >= 1e-15) ? trial[i] : (T)0;   // quirk Q10
                else              trial[i] = (trial[i] > (T)0) ? trial[i] : (T)0;
            }
            f_new = fun_single(ev, P, bsum, trial);
            if (!not_finite(f_new) && f_new <= f_cur - c_ls * step * dd) {
                PMF_EW x[i] = trial[i];
                f_x = f_new;
                break;
            }
            nfeval++;                                              // quirk Q3: failed trials only
            if (nfeval >= maxnfeval) { leave(1); return; }
            step *= decr;
        }
        f_cur = f_new;                                             // quirk Q2, ref: :328
        gprev_sq = ev.dot(g, g);                                   // ref: :332
        PMF_EW { gp[i] = g[i]; dp[i] = d[i]; }
    }
    leave(2);
}

// The same solver with the line search evaluated from cached predictions.  With limit_step the trial point
// x + alpha d stays feasible for alpha <= max_step (only components that land below 1e-15 are snapped to 0), so
// F_j . trial = F_j . x + alpha F_j . d up to <= 1e-15 |F_j|: per CG iteration the tile is read twice (gradient at
// x with p = T.x stored on the way, then q = T.d) and every Armijo trial costs nnz fused-multiply-adds and logs
// instead of another pass over the tile -- for rows that do not fit in LDS that is 2 gathers per iteration
// instead of ~6.  Arithmetic differs from the direct evaluation by rounding only (p + alpha q vs a fresh dot).
template <class EV, class T, int NC>
__device__ __forceinline__ void cg_row_cached(EV& ev, const RowParams<T>& P, const T (&bsum)[NC], T (&x)[NC],
                                              bool weighted, SolveStats& st)
{
    const T tol = (T)1e-2, decr = (T)0.25, c_ls = (T)0.01;
    const int maxnfeval = 150, max_ls = 20;
    const int maxiter = P.maxupd > 0 ? P.maxupd : 0x7fffffff;
    const T two_l2 = (T)(2. * (double)P.l2);
    T g[NC], d[NC], gp[NC], dp[NC], trial[NC], dummy[NC], gz[NC], bs[NC];
    PMF_EW { gp[i] = (T)0; dp[i] = (T)0; dummy[i] = (T)0; g[i] = (T)0; d[i] = (T)0; bs[i] = bsum[i]; }
    T gprev_sq = (T)0;
    // PARKS (fp64 register engine): the k-vectors that must survive a pass over the tile wait in LDS meanwhile (slots 0 x,
    // 1 g, 2 d, 3 previous g, 4 previous d, 5 the constant term), so that the pass has the architectural registers to itself
    // and the tile need not sit in AGPRs, copied out for each use.  The pass's own contribution comes back in gz and is
    // added afterwards (0 + sum, then base + that: the bits of accumulating onto the base).
    constexpr bool PK = EV::PARKS;

    // f0 = f(x) and the first gradient from ONE pass over the row (both are evaluated at x; the reference computes
    // f0 first, ref: src/nonnegcg.c:191, and returns before the gradient if it is not finite -- same outcome), keeping
    // p = T.x on the way
    ev.set_point(x);
    T reg = ev.dot(bs, x);
    reg += P.l2 * ev.dot(x, x);
    PMF_EW gz[i] = (T)0;
    if constexpr (PK) { ev.park(0, x); ev.park(5, bs); }
    T f_cur = reg - (T)ev.template eval<true, true>((T)-1, gz, ev.pbuf) * P.w;
    if constexpr (PK) { ev.unpark(0, x); ev.unpark(5, bs); }
    if (!weighted) {
        PMF_EW {
            g[i] = fma_t(two_l2, x[i], bs[i]);
            g[i] = g[i] + gz[i];
        }
    } else {
        PMF_EW {
            g[i] = gz[i] * P.w;
            g[i] = g[i] + bs[i];
            g[i] = fma_t(two_l2, x[i], g[i]);
        }
    }
    T f_new = (T)0;
    T f_x = f_cur;            // f at the current x (f_cur may be a refused point's value, quirk Q2)
    int nfeval = 1;
    int it = 0;
    auto leave = [&](int rc) { st.niter = it; st.nfeval = nfeval; st.rc = rc; };
    if (not_finite(f_cur)) { leave(3); return; }
    const bool prune = P.x_pos != 0;   // (this variant runs with limit_step only)

    bool p_current = false;   // the cached p is T.x for the current x, to rounding (advanced by a step the cache could be trusted for)
    auto grad_pass = [&](T (&gg)[NC]) {
        if constexpr (EV::CACHED_GRAD) {
            if (p_current) {
                ev.template eval<false, true, true>((T)-1, gg);
                return;
            }
        }
        ev.template eval<false, true>((T)-1, gg, ev.pbuf);
    };
    for (it = 0; it < maxiter; it++) {
        if (it > 0) {
            // gradient at x.  Register engine: after a trusted step the coefficients x_j / p_j come from the cached
            // p = T.x + alpha T.d -- the backward half of a pass; differs from fresh dot products by rounding only, and p is at
            // most `maxupd` updates away from exact values.  Otherwise the pass recomputes F_j . x and refreshes p.
            ev.set_point(x);
            PMF_EW gz[i] = (T)0;
            if constexpr (PK) { ev.park(0, x); ev.park(5, bs); }
            grad_pass(gz);
            if constexpr (PK) { ev.unpark(0, x); ev.unpark(5, bs); ev.unpark(3, gp); ev.unpark(4, dp); }
            if (!weighted) {
                PMF_EW {
                    g[i] = fma_t(two_l2, x[i], bs[i]);
                    g[i] = g[i] + gz[i];
                }
            } else {
                PMF_EW {
                    g[i] = gz[i] * P.w;
                    g[i] = g[i] + bs[i];
                    g[i] = fma_t(two_l2, x[i], g[i]);
                }
            }
        }
        PMF_EW d[i] = (x[i] <= (T)0 && g[i] >= (T)0) ? (T)0 : -g[i];
        T gg_now = (T)0;   // g . g of this iteration's gradient (FUSED_SUMS: reduced together with theta / beta)
        if (it > 0) {
            T th = (T)0, be = (T)0;
            PMF_EW {
                const bool on = ev.act[i] && !(x[i] <= (T)0);
                th += on ? g[i] * dp[i] : (T)0;
                be += on ? g[i] * (g[i] - gp[i]) : (T)0;
            }
            T theta, beta;
            if constexpr (EV::FUSED_SUMS) {
                T v3[3] = { th, be, (T)0 };
                PMF_EW v3[2] = ev.act[i] ? fma_t(g[i], g[i], v3[2]) : v3[2];
                ev.template rsum_n<3>(v3);
                theta = v3[0]; beta = v3[1]; gg_now = v3[2];
            } else {
                theta = ev.rsum(th); beta = ev.rsum(be);
            }
            theta /= gprev_sq;
            beta /= gprev_sq;
            PMF_EW d[i] += (x[i] <= (T)0) ? (T)0 : beta * dp[i] - theta * (g[i] - gp[i]);
        }
        T gd, dd;
        if constexpr (EV::FUSED_SUMS) {
            // g . d and d . d together (and, in the first iteration, g . g with them)
            T v3[3] = { (T)0, (T)0, (T)0 };
            PMF_EW {
                v3[0] = ev.act[i] ? fma_t(g[i], d[i], v3[0]) : v3[0];
                v3[1] = ev.act[i] ? fma_t(d[i], d[i], v3[1]) : v3[1];
                v3[2] = ev.act[i] ? fma_t(g[i], g[i], v3[2]) : v3[2];
            }
            if (it == 0) { ev.template rsum_n<3>(v3); gg_now = v3[2]; }
            else { T v2[2] = { v3[0], v3[1] }; ev.template rsum_n<2>(v2); v3[0] = v2[0]; v3[1] = v2[1]; }
            gd = v3[0]; dd = v3[1];
        } else {
            gd = ev.dot(g, d);
            dd = (T)0;
        }
        if (d_abs((double)gd) <= (double)tol) { leave(0); return; }

        T m = (T)1;
        PMF_EW if (ev.act[i] && d[i] < (T)0) m = (T)d_min((double)m, (double)(-x[i] / d[i]));
        const T max_step = ev.rmin(m);

        // q = T.d (second and last pass over the tile in this iteration)
        ev.set_point(d);
        if constexpr (PK) { ev.park(0, x); ev.park(5, bs); ev.park(1, g); ev.park(2, d); }
        (void)ev.template eval<false, false>((T)0, dummy, ev.qbuf);
        if constexpr (PK) { ev.unpark(0, x); ev.unpark(5, bs); ev.unpark(1, g); ev.unpark(2, d); }

        if constexpr (!EV::FUSED_SUMS) dd = ev.dot(d, d);
        T step = max_step;
        bool accepted = false;
        // (the log-likelihood terms of LS_BATCH consecutive trial steps come from one call: teams of CUs pay one exchange per
        // call; the decisions are those of the one-at-a-time loop)
        constexpr int LSB = EV::LS_BATCH;
        double lsv[LSB];
        bool lst[LSB];
        int ls = 0;
        if (prune && !skip_certain_failures(f_x, f_cur, gd, P.l2 * dd, dd, c_ls, decr, max_ls, maxnfeval, step, ls, nfeval)) { leave(1); return; }
        for (int bpos = 0; ls < max_ls; ls++, bpos++) {
            if constexpr (!EV::FUSED_SUMS) { if (bpos % LSB == 0) ev.logsum_cached_batch(step, decr, lsv, lst); }
            PMF_EW {
                trial[i] = fma_t(step, d[i], x[i]);
                trial[i] = ((double)trial[i] >= 1e-15) ? trial[i] : (T)0;
            }
            T r;
            double lsum_here = 0.0;
            bool trusted = true;
            if constexpr (EV::FUSED_SUMS) {
                // Bsum . trial, trial . trial and the log-likelihood terms of the trial in ONE reduction
                bool bad;
                T v3[3] = { (T)0, (T)0, (T)ev.logsum_cached_lane(step, bad) };
                PMF_EW {
                    v3[0] = ev.act[i] ? fma_t(bs[i], trial[i], v3[0]) : v3[0];
                    v3[1] = ev.act[i] ? fma_t(trial[i], trial[i], v3[1]) : v3[1];
                }
                ev.template rsum_n<3>(v3);
                trusted = __builtin_amdgcn_ballot_w64(bad) == 0;
                r = v3[0];
                r += P.l2 * v3[1];
                lsum_here = ev.combine_scalar(trusted ? (double)v3[2] : __builtin_nan(""));
                trusted = !(lsum_here != lsum_here);
            } else {
                r = ev.dot(bs, trial);
                r += P.l2 * ev.dot(trial, trial);
                lsum_here = lsv[0];
                trusted = lst[0];
#pragma unroll
                for (int j = 1; j < LSB; j++)
                    if (bpos % LSB == j) { lsum_here = lsv[j]; trusted = lst[j]; }
            }
            f_new = r - (T)lsum_here * P.w;
            if (!trusted) f_new = fun_single(ev, P, bs, trial);   // a prediction cancelled to ~0: evaluate at the snapped point
            if (!not_finite(f_new) && f_new <= f_cur - c_ls * step * dd) {
                PMF_EW x[i] = trial[i];
                f_x = f_new;
                accepted = true;
                p_current = trusted;
                if constexpr (EV::CACHED_GRAD) { if (trusted) ev.advance_cached(step); }
                break;
            }
            nfeval++;
            if (nfeval >= maxnfeval) { leave(1); return; }
            step *= decr;
        }
        if (!accepted) p_current = false;   // (x did not move, but the cache may hold a refused trial's history: recompute)
        f_cur = f_new;
        if constexpr (EV::FUSED_SUMS) gprev_sq = gg_now;
        else gprev_sq = ev.dot(g, g);
        if constexpr (PK) { ev.park(3, g); ev.park(4, d); }
        else { PMF_EW { gp[i] = g[i]; dp[i] = d[i]; } }
    }
    leave(2);
}

// ------------------------------------------------------------------------------------------------
// Truncated Newton, lower bound 0                                 ref: src/tnc.c
// ------------------------------------------------------------------------------------------------
// Step-length state of the Gill-Murray safeguarded cubic search (ref: :1822-2154).  Plain uniform
// scalars; in the float build the mixed float/double expressions promote exactly as the C source does.
template <class T> struct Ptc {
    T reltol, abstol, tnytol, fpresn, xbnd, big, rtsmll;
    T u, fu, gu, xmin, fmin, gmin, xw, fw, gw, a, b, oldf, b1, scxbnd, e, step, factor;
    T gtest1, gtest2, tol;
    bool braktd;
};
enum { PTC_OK = 0, PTC_EVAL = 1, PTC_EINVAL = 2, PTC_FAIL = 3 };

template <class T> __device__ __forceinline__ void ptc_clip_step(Ptc<T>& q)
{
    if (q.step >= q.scxbnd) {
        q.step = q.scxbnd;
        q.scxbnd -= (q.reltol * d_abs(q.xbnd) + q.abstol) / (1.0 + q.reltol);
    }
    q.u = q.step;
    if (d_abs(q.step) < q.tol && q.step < 0.0) q.u = -q.tol;
    if (d_abs(q.step) < q.tol && q.step >= 0.0) q.u = q.tol;
}

template <class T> __device__ __forceinline__ int ptc_init(Ptc<T>& q, T eta, T rmu)
{
    if (q.u <= 0.0 || q.xbnd <= q.tnytol || q.gu > 0.0) return PTC_EINVAL;
    if (q.xbnd < q.abstol) q.abstol = q.xbnd;
    q.tol = q.abstol;
    q.a = 0.0; q.xw = 0.0; q.xmin = 0.0;
    q.oldf = q.fu; q.fmin = q.fu; q.fw = q.fu;
    q.gw = q.gu; q.gmin = q.gu;
    q.step = q.u;
    q.factor = 5.0;
    q.braktd = false;
    q.scxbnd = q.xbnd;
    q.b = q.scxbnd + q.reltol * d_abs(q.scxbnd) + q.abstol;
    q.e = q.b + q.b;
    q.b1 = q.b;
    q.gtest1 = -rmu * q.gu;
    q.gtest2 = -eta * q.gu;
    ptc_clip_step(q);
    return PTC_EVAL;
}

template <class T> __device__ __forceinline__ int ptc_iter(Ptc<T>& q)
{
    T r = 0.0, qq = 0.0, s = 0.0, a1, xmidpt, twotol;
    bool skip_update = false;

    if (q.fu <= q.fmin) {
        const T chordu = q.oldf - (q.xmin + q.u) * q.gtest1;
        if (q.fu > chordu) {
            const T chordm = q.oldf - q.xmin * q.gtest1;
            q.gu = -q.gmin;
            T denom = chordm - q.fmin;
            if (d_abs(denom) < 1e-15) {
                denom = 1e-15;
                if (chordm - q.fmin < 0.0) denom = -denom;
            }
            if (q.xmin != 0.0) q.gu = q.gmin * (chordu - q.fu) / denom;
            q.fu = 0.5 * q.u * (q.gmin + q.gu) + q.fmin;
            if (q.fu < q.fmin) q.fu = q.fmin;
        } else {
            q.fw = q.fmin; q.fmin = q.fu;
            q.gw = q.gmin; q.gmin = q.gu;
            q.xmin += q.u;
            q.a -= q.u; q.b -= q.u;
            q.xw = -q.u;
            q.scxbnd -= q.u;
            if (q.gu <= 0.0) q.a = 0.0;
            else { q.b = 0.0; q.braktd = true; }
            q.tol = d_abs(q.xmin) * q.reltol + q.abstol;
            skip_update = true;
        }
    }
    if (!skip_update) {
        if (q.u < 0.0) q.a = q.u;
        else { q.b = q.u; q.braktd = true; }
        q.xw = q.u; q.fw = q.fu; q.gw = q.gu;
    }

    twotol = q.tol + q.tol;
    xmidpt = 0.5 * (q.a + q.b);

    const bool convrg = (d_abs(xmidpt) <= twotol - 0.5 * (q.b - q.a)) ||
                        (d_abs(q.gmin) <= q.gtest2 && q.fmin < q.oldf &&
                         ((d_abs(q.xmin - q.xbnd) > q.tol) || (!q.braktd)));
    if (convrg) {
        if (q.xmin != 0.0) return PTC_OK;
        if (d_abs(q.oldf - q.fw) <= q.fpresn) return PTC_FAIL;
        q.tol = 0.1 * q.tol;
        if (q.tol < q.tnytol) return PTC_FAIL;
        q.reltol = 0.1 * q.reltol;
        q.abstol = 0.1 * q.abstol;
        twotol = 0.1 * twotol;
    }

    bool minimum_found = false;
    if (d_abs(q.e) > q.tol) {
        r = 3.0 * (q.fmin - q.fw) / q.xw + q.gmin + q.gw;
        const T absr = d_abs(r);
        qq = absr;
        if (q.gw != 0.0 && q.gmin != 0.0) {
            const T abgw = d_abs(q.gw), abgmin = d_abs(q.gmin);
            s = d_sqrt(abgmin) * d_sqrt(abgw);
            if (q.gw / abgw * q.gmin > 0.0) {
                if (r >= s || r <= -s) {
                    qq = d_sqrt(d_abs(r + s)) * d_sqrt(d_abs(r - s));
                } else {
                    r = 0.0; qq = 0.0;
                    minimum_found = true;
                }
            } else {
                T sumsq = 1.0, pp = 0.0, scale;
                if (absr >= s) {
                    if (absr > q.rtsmll) pp = absr * q.rtsmll;
                    if (s >= pp) { const T val = s / absr; sumsq = 1.0 + val * val; }
                    scale = absr;
                } else {
                    if (s > q.rtsmll) pp = s * q.rtsmll;
                    if (absr >= pp) { const T val = absr / s; sumsq = 1.0 + val * val; }
                    scale = s;
                }
                sumsq = d_sqrt(sumsq);
                qq = q.big;
                if (scale < q.big / sumsq) qq = scale * sumsq;
            }
        }
        if (!minimum_found) {
            if (q.xw < 0.0) qq = -qq;
            s = q.xw * (q.gmin - r - qq);
            qq = q.gw - q.gmin + qq + qq;
            if (qq > 0.0) s = -s;
            if (qq <= 0.0) qq = -qq;
            r = q.e;
            if (q.b1 != q.step || q.braktd) q.e = q.step;
        }
    }

    a1 = q.a;
    q.b1 = q.b;
    q.step = xmidpt;
    if ((!q.braktd) || ((q.a == 0.0 && q.xw < 0.0) || (q.b == 0.0 && q.xw > 0.0))) {
        if (q.braktd) {
            const T d1 = q.xw;
            T d2 = q.a;
            if (q.a == 0.0) d2 = q.b;
            q.u = -d1 / d2;
            q.step = 5.0 * d2 * (0.1 + 1.0 / q.u) / 11.0;
            if (q.u < 1.0) q.step = 0.5 * d2 * d_sqrt(q.u);
        } else {
            q.step = -q.factor * q.xw;
            if (q.step > q.scxbnd) q.step = q.scxbnd;
            if (q.step != q.scxbnd) q.factor = 5.0 * q.factor;
        }
        if (q.step <= 0.0) a1 = q.step;
        if (q.step > 0.0) q.b1 = q.step;
    }

    if (d_abs(s) <= d_abs(0.5 * qq * r) || s <= qq * a1 || s >= qq * q.b1) {
        q.e = q.b - q.a;
    } else {
        q.step = s / qq;
        if (q.step - q.a < twotol || q.b - q.step < twotol) {
            if (xmidpt <= 0.0) q.step = -q.tol;
            else q.step = q.tol;
        }
    }
    ptc_clip_step(q);
    return PTC_EVAL;
}

enum { T_LOCALMINIMUM = 0, T_FCONVERGED = 1, T_XCONVERGED = 2, T_MAXFUN = 3, T_LSFAIL = 4, T_NOPROGRESS = 6 };
enum { LS_OK = 0, LS_MAXFUN = 1, LS_FAIL = 2 };

// All of TNC's working vectors for one row, in registers.
template <class T, int NC> struct TncState {
    T xscale[NC], xoffset[NC];
    T oldg[NC], g[NC], diagb[NC], pk[NC], sk[NC], yk[NC], sr[NC], yr[NC];
    T r[NC], v[NC], zk[NC], emat[NC], gv[NC];
    T gfull[NC];
    int pivot[NC];
    int nfeval, maxnfeval;
};

template <class T, int NC, class EV> struct Tnc {
    using ST = TncState<T, NC>;
    static constexpr T EPSV = Eps<T>::v;

    static __device__ __forceinline__ void project(const ST& s, T (&v)[NC])      // ref: :1015-1023
    {
        PMF_EW if (s.pivot[i] != 0) v[i] = (T)0.0;
    }
    static __device__ __forceinline__ void unscale_clamp(const ST& s, T (&x)[NC]) // ref: :482-489, :466-479 (Q9)
    {
        PMF_EW {
            x[i] = x[i] * s.xscale[i] + s.xoffset[i];
            x[i] = (x[i] < (T)0.) ? (T)0. : x[i];
        }
    }
    static __device__ __forceinline__ void scaleg(const ST& s, T (&g)[NC], T fscale) // ref: :504-510
    {
        PMF_EW g[i] *= s.xscale[i] * fscale;
    }

    // ref: :1533-1575 with gamma = 1
    static __device__ __forceinline__ void ssbfgs(const T (&sj)[NC], const T (&hv)[NC], const T (&hy)[NC], T ys, T yhy,
                                                  T vs, T vhy, T (&out)[NC])
    {
        const T gamma = 1.0;
        T delta, beta;
        if (ys == 0.0) { delta = 0.0; beta = 0.0; }
        else {
            delta = (gamma * yhy / ys + 1.0) * vs / ys - gamma * vhy / ys;
            beta = -gamma * vs / ys;
        }
        PMF_EW out[i] = gamma * hv[i] + delta * sj[i] + beta * hy[i];
    }

    // ref: :1444-1528.  The reference recomputes everything on every call; within one tnc_direction the preconditioner's
    // inputs (diagb, sk, yk, sr, yr, yksk, yrsr) do not change, so what does not depend on the vector being solved for -- 1 / diagb,
    // H yk (after its own BFGS update), yk.H yk and yr.H yr -- is computed ONCE per direction (msolve_prepare) with the very
    // operations the reference applies, in its order: same bits, five dot products, two ssbfgs scalar blocks and a vector
    // division fewer per CG iteration.
    struct MsolveInv { T rd[NC], hyk[NC]; T ykhyk, yrhyr; };
    static __device__ __forceinline__ void msolve_prepare(const EV& ev, const ST& s, MsolveInv& inv, bool upd1, T yksk, T yrsr, bool lreset)
    {
        (void)yksk;
        if (upd1) return;
        PMF_EW {
            inv.rd[i] = 1.0 / s.diagb[i];
            inv.hyk[i] = s.yk[i] * inv.rd[i];
        }
        if (lreset) {
            inv.ykhyk = ev.dot(s.yk, inv.hyk);
            return;
        }
        T hyr[NC];
        PMF_EW hyr[i] = s.yr[i] * inv.rd[i];
        inv.yrhyr = ev.dot(s.yr, hyr);
        const T yksr = ev.dot(s.yk, s.sr);
        const T ykhyr = ev.dot(s.yk, hyr);
        ssbfgs(s.sr, inv.hyk, hyr, yrsr, inv.yrhyr, yksr, ykhyr, inv.hyk);
        inv.ykhyk = ev.dot(inv.hyk, s.yk);
    }
    static __device__ __forceinline__ void msolve(const EV& ev, const ST& s, const MsolveInv& inv, const T (&g)[NC], T (&y)[NC], bool upd1,
                                                  T yksk, T yrsr, bool lreset)
    {
        if (upd1) {
            PMF_EW y[i] = g[i] / s.diagb[i];
            return;
        }
        const T gsk = ev.dot(g, s.sk);
        T hg[NC];
        PMF_EW hg[i] = g[i] * inv.rd[i];
        if (lreset) {
            const T ghyk = ev.dot(g, inv.hyk);
            ssbfgs(s.sk, hg, inv.hyk, yksk, inv.ykhyk, gsk, ghyk, y);
        } else {
            T hyr[NC];
            PMF_EW hyr[i] = s.yr[i] * inv.rd[i];
            const T gsr = ev.dot(g, s.sr);
            const T ghyr = ev.dot(g, hyr);
            const T ghyk = ev.dot(inv.hyk, g);
            ssbfgs(s.sr, hg, hyr, yrsr, inv.yrhyr, gsr, ghyr, hg);
            ssbfgs(s.sk, hg, inv.hyk, yksk, inv.ykhyk, gsk, ghyk, y);
        }
    }

    // ref: :1580-1658
    static __device__ __forceinline__ void init_precond(const EV& ev, ST& s, bool lreset, T yksk, T yrsr, bool upd1)
    {
        if (upd1) {
            PMF_EW s.emat[i] = s.diagb[i];
            return;
        }
        T bsk[NC];
        if (lreset) {
            PMF_EW bsk[i] = s.diagb[i] * s.sk[i];
            T sds = ev.dot(s.sk, bsk);
            if (yksk == 0.0) yksk = 1.0;
            if (sds == 0.0) sds = 1.0;
            PMF_EW {
                const T td = s.diagb[i];
                s.emat[i] = td - td * td * s.sk[i] * s.sk[i] / sds + s.yk[i] * s.yk[i] / yksk;
            }
        } else {
            PMF_EW bsk[i] = s.diagb[i] * s.sr[i];
            T sds = ev.dot(s.sr, bsk);
            const T srds = ev.dot(s.sk, bsk);
            const T yrsk = ev.dot(s.yr, s.sk);
            if (yrsr == 0.0) yrsr = 1.0;
            if (sds == 0.0) sds = 1.0;
            PMF_EW {
                const T td = s.diagb[i];
                bsk[i] = td * s.sk[i] - bsk[i] * srds / sds + s.yr[i] * yrsk / yrsr;
                s.emat[i] = td - td * td * s.sr[i] * s.sr[i] / sds + s.yr[i] * s.yr[i] / yrsr;
            }
            sds = ev.dot(s.sk, bsk);
            if (yksk == 0.0) yksk = 1.0;
            if (sds == 0.0) sds = 1.0;
            PMF_EW s.emat[i] -= bsk[i] * bsk[i] / sds + s.yk[i] * s.yk[i] / yksk;
        }
    }

    // ref: :1388-1435, one fused evaluation
    static __device__ __forceinline__ void hess_vec(EV& ev, const RowParams<T>& P, const T (&bsum)[NC], ST& s,
                                                    const T (&x)[NC], T fscale, T accuracy, T xnorm)
    {
        const T delta = accuracy * (xnorm + 1.0);
        T xv[NC];
        PMF_EW xv[i] = x[i] + delta * s.v[i];
        unscale_clamp(s, xv);
        (void)fun_and_grad(ev, P, bsum, xv, s.gv);
        scaleg(s, s.gv, fscale);
        const T dinv = 1.0 / delta;
        PMF_EW s.gv[i] = (s.gv[i] - s.g[i]) * dinv;
    }

    // ref: :1162-1341
    static __device__ __forceinline__ void direction(EV& ev, const RowParams<T>& P, const T (&bsum)[NC], ST& s,
                                                     const T (&x)[NC], int maxCGit, bool upd1, T yksk, T yrsr,
                                                     bool lreset, T fscale, T accuracy, T gnorm, T xnorm)
    {
        T (&zsol)[NC] = s.pk;
        if (maxCGit == 0) {
            PMF_EW zsol[i] = -s.g[i];
            project(s, zsol);
            return;
        }
        const T rhsnrm = gnorm, tol = 1e-12;
        T qold = 0.0, rzold = 0.0;
        init_precond(ev, s, lreset, yksk, yrsr, upd1);
        MsolveInv inv;
        msolve_prepare(ev, s, inv, upd1, yksk, yrsr, lreset);
        PMF_EW { s.r[i] = -s.g[i]; s.v[i] = 0.0; zsol[i] = 0.0; }

        for (int it = 0; it < maxCGit; it++) {
            project(s, s.r);
            msolve(ev, s, inv, s.r, s.zk, upd1, yksk, yrsr, lreset);
            project(s, s.zk);
            const T rz = ev.dot(s.r, s.zk);
            if ((rz / rhsnrm < tol) || (s.nfeval >= (s.maxnfeval - 1))) {
                if (it == 0) {
                    PMF_EW zsol[i] = -s.g[i];
                    project(s, zsol);
                }
                break;
            }
            const T beta = (it == 0) ? (T)0.0 : rz / rzold;
            PMF_EW s.v[i] = s.zk[i] + beta * s.v[i];
            project(s, s.v);

            hess_vec(ev, P, bsum, s, x, fscale, accuracy, xnorm);
            s.nfeval++;
            project(s, s.gv);

            const T vgv = ev.dot(s.v, s.gv);
            if (vgv / rhsnrm < tol) {
                if (it == 0) {
                    msolve(ev, s, inv, s.g, zsol, upd1, yksk, yrsr, lreset);
                    PMF_EW zsol[i] = -zsol[i];
                    project(s, zsol);
                }
                break;
            }
            {   // diagonalScaling, ref: :1347-1362
                const T vr = 1.0 / ev.dot(s.v, s.r);
                const T ivgv = 1.0 / vgv;                                 // (the reference takes v.gv a second time: the same sum)
                PMF_EW {
                    s.emat[i] += -s.r[i] * s.r[i] * vr + s.gv[i] * s.gv[i] * ivgv;
                    s.emat[i] = ((double)s.emat[i] <= 1e-6) ? (T)1. : s.emat[i];
                }
            }
            const T alpha = rz / vgv;
            PMF_EW {
                zsol[i] = fma_t(alpha, s.v[i], zsol[i]);
                s.r[i] = fma_t(-alpha, s.gv[i], s.r[i]);
            }
            const T gtp = ev.dot(zsol, s.g);
            const T pr = ev.dot(s.r, zsol);
            const T qnew = (gtp + pr) * 0.5;
            const T qtest = (it + 1) * (1.0 - qold / qnew);
            if (qtest <= 0.5) break;
            if (gtp > 0.0) {
                PMF_EW zsol[i] = fma_t(-alpha, s.v[i], zsol[i]);
                break;
            }
            qold = qnew;
            rzold = rz;
        }
        PMF_EW s.diagb[i] = s.emat[i];
    }

    // ref: :1664-1813.  r / v / zk double as temp / tempgfull / newgfull exactly as the reference's
    // buffer carving makes them (they are dead between tnc_direction and here).
    static __device__ __forceinline__ int linesearch(EV& ev, const RowParams<T>& P, const T (&bsum)[NC], ST& s,
                                                     T fscale, T eta, T ftol, T xbnd, T (&x)[NC], T& f, T& alpha)
    {
        const int maxlsit = 64;
        Ptc<T> q;
        T (&temp)[NC] = s.r;
        T (&tempg)[NC] = s.v;
        T (&newg)[NC] = s.zk;
        const T (&p)[NC] = s.pk;

        PMF_EW temp[i] = s.gfull[i];
        scaleg(s, temp, fscale);
        q.gu = ev.dot(temp, p);

        PMF_EW temp[i] = x[i];
        project(s, temp);
        const T xnorm = ev.nrm2(temp);

        const T rteps = d_sqrt(EPSV);
        const T pe = ev.nrm2(p) + EPSV;
        q.reltol = rteps * (xnorm + 1.0) / pe;
        q.abstol = -EPSV * (1.0 + d_abs(f)) / (q.gu - EPSV);
        q.tnytol = EPSV * (xnorm + 1.0) / pe;
        q.rtsmll = EPSV;
        q.big = 1.0 / (EPSV * EPSV);
        q.fpresn = ftol;
        q.xbnd = xbnd;
        q.u = alpha;
        q.xmin = alpha;
        q.fu = f;
        q.fmin = f;
        const T rmu = 1e-4;

        int itcnt = 0;
        int itest = ptc_init(q, eta, rmu);
        while (itest == PTC_EVAL) {
            if ((++itcnt > maxlsit) || (s.nfeval >= s.maxnfeval)) break;
            const T ualpha = q.xmin + q.u;
            PMF_EW temp[i] = x[i] + ualpha * p[i];
            unscale_clamp(s, temp);
            q.fu = fun_and_grad(ev, P, bsum, temp, tempg);
            s.nfeval++;
            q.fu *= fscale;
            PMF_EW temp[i] = tempg[i];
            scaleg(s, temp, fscale);
            q.gu = ev.dot(temp, p);
            itest = ptc_iter(q);
            if (q.xmin == ualpha) { PMF_EW newg[i] = tempg[i]; }
        }
        alpha = q.xmin;
        if (itest == PTC_OK) {
            f = q.fmin;
            PMF_EW x[i] = fma_t(alpha, p[i], x[i]);
            PMF_EW s.gfull[i] = newg[i];
            return LS_OK;
        }
        if (itcnt > maxlsit) return LS_FAIL;
        if (itest != PTC_EVAL) return LS_FAIL;
        return LS_MAXFUN;
    }

    // ref: tnc :251-463 + tnc_minimize :554-993 with tncg_iteration's arguments (src/poismf.c:383-391)
    static __device__ __forceinline__ int minimize(EV& ev, const RowParams<T>& P, const T (&bsum)[NC], T (&x)[NC], SolveStats& st)
    {
        ST s;
        s.nfeval = 0;
        s.maxnfeval = P.maxupd;
        int maxCGit = P.max_cg_it;
        const int n = ev.k;

        PMF_EW x[i] = (x[i] < (T)0.) ? (T)0. : x[i];                      // coercex, ref: :323
        if (s.maxnfeval < 1) { st.rc = T_MAXFUN; return T_MAXFUN; }
        T f = fun_and_grad(ev, P, bsum, x, s.gfull);                      // ref: :341
        s.nfeval++;
        PMF_EW {                                                          // ref: :383-399 (Q9)
            s.xscale[i] = 1.0 + d_abs(x[i]);
            s.xoffset[i] = x[i];
        }
        T fscale = 1.0;
        const T rteps = d_sqrt(EPSV);
        T stepmx = 10.;
        const T eta = 0.25, rescale = 1.3, fmin_est = 0., ftol = 1e-4;
        T accuracy = 0., xtol = -1., pgtol = -1.;
        if (stepmx < rteps * 10.0) stepmx = 1.0e1;
        if (maxCGit > n) maxCGit = n;
        if (accuracy <= EPSV) accuracy = rteps;
        if (pgtol < 0.0) pgtol = 1e-2 * d_sqrt(accuracy);
        if (xtol < 0.0) xtol = rteps;

        T difnew = 0.0, epsred = 0.05, difold, oldf, oldgtp, xnorm, gnorm, ustpmax, spe;
        T fLastReset, fLastConstraint, yrsr = 0.0, yksk = 0.0, alpha = 0.0;
        bool upd1 = true, newcon = true, lreset = false, remcon;
        int icycle = n - 1, rc;
        int niter = 0;
        T temp[NC];

        PMF_EW if (s.xscale[i] > 0.0) x[i] = (x[i] - s.xoffset[i]) / s.xscale[i];     // scalex, ref: :492-501
        f *= fscale;
        PMF_EW {                                                          // setConstraints, low = 0, ref: :513-545
            if (s.xscale[i] == 0.0) s.pivot[i] = 2;
            else if (x[i] * s.xscale[i] + s.xoffset[i] - (T)0. <= EPSV * 10.0 * (d_abs((T)0.) + 1.0)) s.pivot[i] = -1;
            else s.pivot[i] = 0;
        }
        PMF_EW s.g[i] = s.gfull[i];
        scaleg(s, s.g, fscale);
        PMF_EW if (-s.pivot[i] * s.g[i] < 0.0) s.pivot[i] = 0;           // ref: :670-674
        project(s, s.g);
        gnorm = ev.nrm2(s.g);
        fLastConstraint = f;
        fLastReset = f;
        PMF_EW { s.diagb[i] = 1.0; s.sk[i] = 0; s.yk[i] = 0; s.sr[i] = 0; s.yr[i] = 0; }

        for (;;) {
            T newscale = ev.nrm2(s.g);                                    // (taken twice in the reference, ref: :700, :720: the same sum)
            if (newscale <= pgtol * fscale) { rc = T_LOCALMINIMUM; break; }   // ref: :700-712
            if (s.nfeval >= s.maxnfeval) { rc = T_MAXFUN; break; }
                                                                          // ref: :720-746
            if ((newscale > EPSV) && (d_abs(d_log10(newscale)) > rescale)) {
                newscale = 1.0 / newscale;
                f *= newscale; fscale *= newscale; gnorm *= newscale;
                fLastConstraint *= newscale; fLastReset *= newscale; difnew *= newscale;
                PMF_EW s.g[i] *= newscale;
                PMF_EW s.diagb[i] = 1.0;
                upd1 = true; icycle = n - 1; newcon = true;
            }

            PMF_EW temp[i] = x[i];
            project(s, temp);
            xnorm = ev.nrm2(temp);
            const int oldnfeval = s.nfeval;

            direction(ev, P, bsum, s, x, maxCGit, upd1, yksk, yrsr, lreset, fscale, accuracy, gnorm, xnorm);

            if (!newcon) {                                                // ref: :770-785
                if (!lreset) {
                    PMF_EW { s.sr[i] = s.sr[i] + s.sk[i]; s.yr[i] = s.yr[i] + s.yk[i]; }
                    icycle++;
                } else {
                    PMF_EW { s.sr[i] = s.sk[i]; s.yr[i] = s.yk[i]; }
                    fLastReset = f;
                    icycle = 1;
                }
            }
            PMF_EW s.oldg[i] = s.g[i];
            oldf = f;
            oldgtp = ev.dot(s.pk, s.g);

            ustpmax = stepmx / (ev.nrm2(s.pk) + EPSV);
            {                                                             // stepMax, low = 0, up = inf, ref: :1041-1067
                // sequential in the reference (each accepted bound tightens `step` for the next
                // comparison); the final value is the minimum of ustpmax and the admissible ratios,
                // which a min-reduction reproduces exactly because t/dir is evaluated the same way
                T m = ustpmax;
                PMF_EW {
                    if (ev.act[i] && (s.pivot[i] == 0) && (s.pk[i] < 0.0)) {
                        const T t = ((T)0. - s.xoffset[i]) / s.xscale[i] - x[i];
                        if (t > ustpmax * s.pk[i]) m = (T)d_min((double)m, (double)(t / s.pk[i]));
                    }
                }
                spe = ev.rmin(m);
            }

            if (spe > 0.0) {
                {                                                         // initialStep, ref: :1368-1383
                    const T d = d_abs(f - fmin_est / fscale);
                    alpha = 1.0;
                    if (d * 2.0 <= -oldgtp && d >= EPSV) alpha = d * -2.0 / oldgtp;
                    if (alpha >= spe) alpha = spe;
                }
                const int lsrc = linesearch(ev, P, bsum, s, fscale, eta, ftol, spe, x, f, alpha);
                if (lsrc == LS_FAIL) { rc = T_LSFAIL; break; }
                if (alpha >= 0.9 * ustpmax) stepmx *= 1e2;
                if (alpha - spe >= -EPSV * 10.0) newcon = true;
                else {
                    if (lsrc != LS_OK) { rc = (lsrc == LS_MAXFUN) ? T_MAXFUN : T_LSFAIL; break; }
                    newcon = false;
                }
            } else {
                newcon = true;
            }

            if (newcon) {                                                 // addConstraint, ref: :1072-1108
                bool added = false;
                PMF_EW {
                    if (ev.act[i] && (s.pivot[i] == 0) && (s.pk[i] < 0.0)) {
                        const T tolc = EPSV * 10.0 * (d_abs((T)0.) + 1.0);
                        if (x[i] * s.xscale[i] + s.xoffset[i] - (T)0. <= tolc) {
                            s.pivot[i] = -1;
                            x[i] = ((T)0. - s.xoffset[i]) / s.xscale[i];
                            added = true;
                        }
                    }
                }
                added = ev.rmax((int)added) != 0;
                if (!added && s.nfeval == oldnfeval) { rc = T_NOPROGRESS; break; }
                fLastConstraint = f;
            }
            niter++;                                                      // (tnc_minimize's own iteration count)

            difold = difnew;
            difnew = oldf - f;
            if (icycle == 1) {
                if (difnew > difold * 2.0) epsred += epsred;
                if (difnew < difold * 0.5) epsred *= 0.5;
            }

            PMF_EW s.g[i] = s.gfull[i];
            scaleg(s, s.g, fscale);
            PMF_EW temp[i] = s.g[i];
            project(s, temp);
            gnorm = ev.nrm2(temp);

            remcon = false;                                               // removeConstraint, ref: :1113-1153
            if (!(((fLastConstraint - f) <= (oldgtp * -0.5)) && (gnorm > pgtol * fscale))) {
                // first index attaining the most negative multiplier (strict `<` scan in the reference)
                T best = 0.0;
                int besti = 0x7fffffff;
                PMF_EW {
                    if (ev.act[i] && s.pivot[i] != 2) {
                        const T t = -s.pivot[i] * s.g[i];
                        if (t < best) { best = t; besti = ev.elem[i]; }
                    }
                }
                const T cmax = ev.rmin(best);
                if (cmax < 0.0) {
                    const int imax = ev.rmin((best == cmax) ? besti : 0x7fffffff);
                    PMF_EW if (ev.elem[i] == imax) s.pivot[i] = 0;
                    remcon = true;
                }
            }
            if (remcon) {
                PMF_EW temp[i] = s.g[i];
                project(s, temp);
                gnorm = ev.nrm2(temp);
                fLastConstraint = f;
            }

            if (!remcon && !newcon) {                                     // ref: :909-929
                if (d_abs(difnew) <= ftol * fscale) { rc = T_FCONVERGED; break; }
                if (alpha * ev.nrm2(s.pk) <= xtol) { rc = T_XCONVERGED; break; }
            }
            project(s, s.g);

            if (!newcon) {                                                // ref: :940-962
                PMF_EW {
                    s.yk[i] = s.g[i] - s.oldg[i];
                    s.sk[i] = alpha * s.pk[i];
                }
                yksk = ev.dot(s.yk, s.sk);
                if (icycle == (n - 1) || difnew < epsred * (fLastReset - f)) lreset = true;
                else {
                    yrsr = ev.dot(s.yr, s.sr);
                    lreset = (yrsr <= 0.0);
                }
                upd1 = false;
            }
        }

        unscale_clamp(s, x);                                              // ref: :971-972
        st.niter = niter; st.nfeval = s.nfeval; st.rc = rc;
        return rc;
    }
};

}  // namespace pmf
