/*
 * poismf_oracle.c -- CPU restatement of poismf's alternating factor-update hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker.  The shipped path is the HIP library (poismf_amd/csrc), which has no CPU fallback.
 *
 * Parity status: PINNED.  Every exported function below is checked in tests/test_oracle_vs_ref.py
 * against the real reference compiled in place from /root/reference/src (oracle/Makefile `ref`
 * target -> oracle/_ref/), and against the golden vectors in tests/golden/ that were minted from
 * that same compiled reference by scripts/make_golden.py.
 *
 * This is an independent restatement written from the mathematics and the quirk list of
 * SURVEY.md (Q1-Q13); each function cites the reference lines it follows.  All "ref:" citations
 * are relative to /root/reference/.  Compile with -DUSE_FLOAT for the fp32 variant (the reference
 * builds its float flavour the same way, ref: src/poismf.h:91-109).
 *
 * Arithmetic conventions (so that results are reproducible on any x86-64 host):
 *   - k-length dot products / axpys are plain left-to-right loops (the reference delegates them to
 *     whichever BLAS its wrapper forwards to, so the summation order is not part of its contract);
 *   - compiled with -ffp-contract=off;
 *   - wherever the reference mixes `double` literals / libm double functions into real_t
 *     expressions (which matters for the float build), the same promotions are kept.
 */
#include <float.h>
#include <math.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_thread_num(void) { return 0; }
#endif

#ifdef USE_FLOAT
typedef float real_t;
#define R_EPS FLT_EPSILON
#else
typedef double real_t;
#define R_EPS DBL_EPSILON
#endif
typedef size_t sparse_ix;

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* k-length vector helpers (the reference's cblas_t{dot,axpy,scal,nrm2}, ref: src/poismf.h:141) */
/* ------------------------------------------------------------------------------------------ */
#ifdef ORACLE_CBLAS
/* Test-only flavour (liboracle_blas_*.so): route the k-length operations through the very BLAS the
   compiled reference uses (SciPy's OpenBLAS) so that restatement and reference follow the same
   summation order; what remains is the control flow, which must then agree almost bit for bit. */
#ifdef USE_FLOAT
#define BL(name) scipy_cblas_s##name
#else
#define BL(name) scipy_cblas_d##name
#endif
extern real_t BL(dot)(const int n, const real_t *x, const int incx, const real_t *y, const int incy);
extern void BL(axpy)(const int n, const real_t a, const real_t *x, const int incx, real_t *y, const int incy);
extern void BL(scal)(const int n, const real_t a, real_t *x, const int incx);
extern real_t BL(nrm2)(const int n, const real_t *x, const int incx);
static inline real_t vdot(int n, const real_t *x, const real_t *y) { return BL(dot)(n, x, 1, y, 1); }
static inline void vaxpy(int n, real_t a, const real_t *x, real_t *y) { BL(axpy)(n, a, x, 1, y, 1); }
static inline void vscal(int n, real_t a, real_t *x) { BL(scal)(n, a, x, 1); }
static inline real_t vnrm2(int n, const real_t *x) { return BL(nrm2)(n, x, 1); }
#else
static inline real_t vdot(int n, const real_t *x, const real_t *y)
{
    real_t s = 0;
    for (int i = 0; i < n; i++) s += x[i] * y[i];
    return s;
}
#ifdef ORACLE_FMA_AXPY
/* Flavour liboracle_fma_*.so: y += a x with ONE rounding per element, as every BLAS with fused multiply-add does
   (the SciPy OpenBLAS the compiled reference links on x86-64; the GPU kernels too).  It matters in the float build of
   minimize_nonneg_cg: a step limited by max_step = -x_i / d_i lands coordinate i on x_i + step d_i, which is exactly 0
   most of the time with two roundings but the division's rounding residual (~1e-9, above the 1e-15 snap of
   ref src/nonnegcg.c:301-303) with one -- and a coordinate left at 1e-9 limits the NEXT step to nothing.  The compiled
   reference shows exactly that behaviour on 1000-nonzero rows (scripts/probes/probe_cg32_long.py). */
static inline void vaxpy(int n, real_t a, const real_t *x, real_t *y)
{
#ifdef USE_FLOAT
    for (int i = 0; i < n; i++) y[i] = __builtin_fmaf(a, x[i], y[i]);
#else
    for (int i = 0; i < n; i++) y[i] = __builtin_fma(a, x[i], y[i]);
#endif
}
#else
static inline void vaxpy(int n, real_t a, const real_t *x, real_t *y)
{
    for (int i = 0; i < n; i++) y[i] += a * x[i];
}
#endif
static inline void vscal(int n, real_t a, real_t *x)
{
    for (int i = 0; i < n; i++) x[i] *= a;
}
/* Euclidean norm.  Reference BLAS nrm2 uses a scaled algorithm; for the magnitudes met here a
   plain sqrt of the sum of squares differs by rounding only. */
static inline real_t vnrm2(int n, const real_t *x)
{
    real_t s = 0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    return (real_t)sqrt(s);
}
#endif

/* One row sub-problem: the closure the reference passes as `fdata` (ref: src/poismf.h:121-130). */
typedef struct {
    const real_t *F;       /* opposing factor, [dimF x k] row-major         */
    const real_t *bsum;    /* k-vector: colsum(F) + l1 (or per-row variant) */
    const real_t *xval;    /* this row's nonzero values                     */
    const sparse_ix *xind; /* this row's nonzero column indices             */
    size_t nnz;
    real_t l2, w;
    int k;
} rowprob;

/* ------------------------------------------------------------------------------------------ */
/* Row primitives                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* out = sum_j (x_j / (F_j . a)) F_j          ref: src/poismf.c:126-133 (calc_grad_pgd) */
ORC_API void oracle_calc_grad_pgd(real_t *out, const real_t *a, const real_t *F, const real_t *xval,
                                  const sparse_ix *xind, size_t nnz, int k)
{
    for (int c = 0; c < k; c++) out[c] = 0;
    for (size_t j = 0; j < nnz; j++) {
        const real_t *Fj = F + (size_t)xind[j] * (size_t)k;
        real_t coef = xval[j] / vdot(k, Fj, a);
        vaxpy(k, coef, Fj, out);
    }
}

/* f = bsum.a + l2 (a.a) - w sum_j x_j log(a.F_j)      ref: src/poismf.c:194-208 (calc_fun_single).
   `log` is the double libm function even in the float build; the running sum is a real_t that is
   updated through a double expression (x_j promoted by the double log). */
static real_t row_fun(const rowprob *p, const real_t *a)
{
    int k = p->k;
    real_t reg = vdot(k, p->bsum, a);
    reg += p->l2 * vdot(k, a, a);
    real_t lsum = 0;
    for (size_t j = 0; j < p->nnz; j++)
        lsum += p->xval[j] * log(vdot(k, a, p->F + (size_t)p->xind[j] * (size_t)k));
    return reg - lsum * p->w;
}

/* g = bsum + 2 l2 a - sum_j (x_j/(a.F_j)) F_j         ref: src/poismf.c:210-223 (calc_grad_single) */
static void row_grad(const rowprob *p, const real_t *a, real_t *g)
{
    int k = p->k;
    memcpy(g, p->bsum, sizeof(real_t) * (size_t)k);
    vaxpy(k, (real_t)(2. * p->l2), a, g);
    for (size_t j = 0; j < p->nnz; j++) {
        const real_t *Fj = p->F + (size_t)p->xind[j] * (size_t)k;
        vaxpy(k, -p->xval[j] / vdot(k, a, Fj), Fj, g);
    }
}

/* weighted variant, different accumulation order (quirk Q11)   ref: src/poismf.c:225-240 */
static void row_grad_w(const rowprob *p, const real_t *a, real_t *g)
{
    int k = p->k;
    for (int c = 0; c < k; c++) g[c] = 0;
    for (size_t j = 0; j < p->nnz; j++) {
        const real_t *Fj = p->F + (size_t)p->xind[j] * (size_t)k;
        vaxpy(k, -p->xval[j] / vdot(k, a, Fj), Fj, g);
    }
    vscal(k, p->w, g);
    vaxpy(k, (real_t)1., p->bsum, g);
    vaxpy(k, (real_t)(2. * p->l2), a, g);
}

/* fused f and g for TNC.  Quirk Q4: f omits the l2 term, g carries it.
   ref: src/poismf.c:242-273 (calc_fun_and_grad) */
static void row_fun_grad(const rowprob *p, const real_t *a, real_t *f, real_t *g)
{
    int k = p->k;
    real_t lsum = 0;
    for (int c = 0; c < k; c++) g[c] = 0;
    for (size_t j = 0; j < p->nnz; j++) {
        const real_t *Fj = p->F + (size_t)p->xind[j] * (size_t)k;
        real_t pred = vdot(k, a, Fj);
        vaxpy(k, -p->xval[j] / pred, Fj, g);
        lsum += p->xval[j] * log(pred);
    }
    if (p->w != 1.) vscal(k, p->w, g);
    vaxpy(k, (real_t)1., p->bsum, g);
    real_t reg = vdot(k, p->bsum, a);
    vaxpy(k, (real_t)(2. * p->l2), a, g);
    *f = reg - lsum * p->w;
}

/* ctypes-friendly wrappers for golden level G1 */
static rowprob mk_prob(const real_t *F, const real_t *bsum, const real_t *xval, const sparse_ix *xind,
                       size_t nnz, int k, real_t l2, real_t w)
{
    rowprob p = { F, bsum, xval, xind, nnz, l2, w, k };
    return p;
}
ORC_API real_t oracle_calc_fun_single(const real_t *a, const real_t *F, const real_t *bsum,
                                      const real_t *xval, const sparse_ix *xind, size_t nnz, int k,
                                      real_t l2, real_t w)
{
    rowprob p = mk_prob(F, bsum, xval, xind, nnz, k, l2, w);
    return row_fun(&p, a);
}
ORC_API void oracle_calc_grad_single(real_t *g, const real_t *a, const real_t *F, const real_t *bsum,
                                     const real_t *xval, const sparse_ix *xind, size_t nnz, int k,
                                     real_t l2, real_t w, int weighted_variant)
{
    rowprob p = mk_prob(F, bsum, xval, xind, nnz, k, l2, w);
    if (weighted_variant) row_grad_w(&p, a, g); else row_grad(&p, a, g);
}
ORC_API real_t oracle_calc_fun_and_grad(real_t *g, const real_t *a, const real_t *F, const real_t *bsum,
                                        const real_t *xval, const sparse_ix *xind, size_t nnz, int k,
                                        real_t l2, real_t w)
{
    rowprob p = mk_prob(F, bsum, xval, xind, nnz, k, l2, w);
    real_t f;
    row_fun_grad(&p, a, &f, g);
    return f;
}

/* ------------------------------------------------------------------------------------------ */
/* Non-negative PRP conjugate gradient for one row          ref: src/nonnegcg.c:177-346        */
/* ------------------------------------------------------------------------------------------ */
#define CG_EPS 1e-15 /* ref: src/nonnegcg.c:94 */
static inline bool not_finite(real_t v) { return isnan(v) || isinf(v); }

/* returns the reference's cg_result code (ref: src/nonnegcg.c:136-138); scratch = 5k reals */
static int row_cg(const rowprob *p, real_t *x, bool weighted_grad, real_t tol, size_t maxnfeval,
                  size_t maxiter, real_t decr, real_t c_ls, size_t max_ls, bool limit_step,
                  real_t *scratch, real_t *fun_out, size_t *niter_out, size_t *nfeval_out)
{
    const int n = p->k;
    real_t *gbuf[2] = { scratch, scratch + n };
    real_t *dbuf[2] = { scratch + 2 * n, scratch + 3 * n };
    real_t *trial = scratch + 4 * n;
    int cur = 0; /* ping-pong slot of the current gradient / direction (ref: :204, :335-339) */
    real_t gprev_sq = 0;
    real_t f_cur = row_fun(p, x); /* ref: :191 */
    real_t f_new = 0;
    size_t nfeval = 1, it = 0;
    int rc = 2; /* stop_maxiter */
    if (maxiter == 0) maxiter = INT32_MAX;
    if (maxnfeval == 0) maxnfeval = INT32_MAX;

    if (not_finite(f_cur)) { rc = 3; goto done; } /* ref: :223-226 */

    for (it = 0; it < maxiter; it++) {
        real_t *g = gbuf[cur], *d = dbuf[cur];
        const real_t *gp = gbuf[cur ^ 1], *dp = dbuf[cur ^ 1];

        if (weighted_grad) row_grad_w(p, x, g); else row_grad(p, x, g); /* ref: :231 */

        /* capped steepest descent (ref: :236-239) */
        for (int i = 0; i < n; i++)
            d[i] = (x[i] <= 0. && g[i] >= 0.) ? (real_t)0. : -g[i];

        if (it > 0) { /* ref: :242-261 */
            real_t theta = 0, beta = 0;
            /* the `0.` arms make these conditional expressions double-typed in the float build, as in
               the reference: the real_t product is formed first, then added through a double */
            for (int i = 0; i < n; i++) {
                theta += (x[i] <= 0.) ? 0. : g[i] * dp[i];
                beta += (x[i] <= 0.) ? 0. : g[i] * (g[i] - gp[i]);
            }
            theta /= gprev_sq;
            beta /= gprev_sq;
            for (int i = 0; i < n; i++)
                d[i] += (x[i] <= 0.) ? 0. : beta * dp[i] - theta * (g[i] - gp[i]);
        }

        /* stopping rule on <g,d> (ref: :264-269) */
        real_t gd = vdot(n, g, d);
        if (fabs(gd) <= tol) { rc = 0; goto done; }

        /* largest admissible step (ref: :272-288) */
        real_t max_step;
        if (limit_step) {
            max_step = 1.;
            for (int i = 0; i < n; i++)
                if (d[i] < 0.) max_step = (real_t)fmin(max_step, -x[i] / d[i]);
        } else {
            max_step = 0.;
            for (int i = 0; i < n; i++)
                if (d[i] < 0.) max_step = (real_t)fmax(max_step, -x[i] / d[i]);
            max_step = (real_t)fmin(1., 0.99 * max_step);
        }

        /* backtracking (ref: :295-327).  Quirk Q3: nfeval counts failed trials only.  Quirk Q2: the
           reference's `ls == max_ls + 1` exit is unreachable, so after max_ls failures x is kept,
           f_cur becomes the last *rejected* value and the outer loop simply continues. */
        real_t dd = vdot(n, d, d);
        real_t step = max_step;
        for (size_t ls = 0; ls < max_ls; ls++) {
            memcpy(trial, x, sizeof(real_t) * (size_t)n);
            vaxpy(n, step, d, trial);
            if (limit_step) {
                for (int i = 0; i < n; i++) trial[i] = (trial[i] >= CG_EPS) ? trial[i] : (real_t)0.;
            } else {
                for (int i = 0; i < n; i++) trial[i] = (trial[i] > 0.) ? trial[i] : (real_t)0.;
            }
            f_new = row_fun(p, trial);
            if (!not_finite(f_new)) {
                if (f_new <= f_cur - c_ls * step * dd) {
                    memcpy(x, trial, sizeof(real_t) * (size_t)n);
                    break;
                }
            }
            nfeval++;
            if (nfeval >= maxnfeval) { rc = 1; goto done; }
            step *= decr;
        }
        f_cur = f_new; /* ref: :328 */

        gprev_sq = vdot(n, g, g); /* ref: :332, norm of the FULL gradient */
        cur ^= 1;
    }

done:
    *fun_out = f_cur;
    *niter_out = it;
    *nfeval_out = nfeval;
    return rc;
}

/* G2 entry point: one row through the CG solver with cg_iteration's constants
   (ref: src/poismf.c:315-320: tol 1e-2, maxnfeval 150, decr 0.25, c 0.01, max_ls 20). */
ORC_API int oracle_cg_row(real_t *x, const real_t *F, const real_t *bsum, const real_t *xval,
                          const sparse_ix *xind, size_t nnz, int k, real_t l2, real_t w,
                          size_t maxupd, int limit_step, real_t *fun_out, size_t *niter_out,
                          size_t *nfeval_out)
{
    rowprob p = mk_prob(F, bsum, xval, xind, nnz, k, l2, w);
    real_t *scratch = (real_t *)malloc(sizeof(real_t) * 5 * (size_t)k);
    int rc = row_cg(&p, x, w != 1., (real_t)1e-2, 150, maxupd, (real_t)0.25, (real_t)0.01, 20,
                    limit_step != 0, scratch, fun_out, niter_out, nfeval_out);
    free(scratch);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Truncated Newton with the lower bound hard-wired to 0     ref: src/tnc.c                    */
/* ------------------------------------------------------------------------------------------ */
/* Return codes follow ref: src/tnc.h:68-83 */
enum { T_LOCALMINIMUM = 0, T_FCONVERGED = 1, T_XCONVERGED = 2, T_MAXFUN = 3, T_LSFAIL = 4,
       T_CONSTANT = 5, T_NOPROGRESS = 6 };

typedef struct {
    const rowprob *prob;
    int n;
    /* 2k + 9k + 5k + 3k = 19k reals, as the reference carves them (ref: src/tnc.c:361-376,
       :622-631, :1220-1225; the three `aux` vectors are the overlapping temporaries of
       :1605, :1480-1483, :1409 and linearSearch's :1700-1703, which are never live together) */
    real_t *xscale, *xoffset;
    real_t *oldg, *g, *temp, *diagb, *pk, *sk, *yk, *sr, *yr;
    real_t *r, *v, *zk, *emat, *gv;
    real_t *aux0, *aux1, *aux2;
    int *pivot;
    int nfeval, maxnfeval;
} tnc_ws;

static void t_project(int n, real_t *v, const int *pivot) /* ref: :1015-1023 */
{
    for (int i = 0; i < n; i++) if (pivot[i] != 0) v[i] = 0.0;
}
static void t_clamp0(int n, real_t *x) /* coercex: bounds ignored, clamp at 0 (Q9) ref: :466-479 */
{
    for (int i = 0; i < n; i++) x[i] = (x[i] < 0.) ? (real_t)0. : x[i];
}
static void t_unscale(int n, real_t *x, const real_t *xs, const real_t *xo) /* ref: :482-489 */
{
    for (int i = 0; i < n; i++) x[i] = x[i] * xs[i] + xo[i];
}
static void t_scaleg(int n, real_t *g, const real_t *xs, real_t fscale) /* ref: :504-510 */
{
    for (int i = 0; i < n; i++) g[i] *= xs[i] * fscale;
}

/* Self-scaled BFGS update with gamma = 1 (ref: :1533-1575; out may alias hv) */
static void t_ssbfgs(int n, const real_t *sj, const real_t *hv, const real_t *hy, real_t ys,
                     real_t yhy, real_t vs, real_t vhy, real_t *out)
{
    const real_t gamma = 1.0;
    real_t delta, beta;
    if (ys == 0.0) { delta = 0.0; beta = 0.0; }
    else {
        delta = (gamma * yhy / ys + 1.0) * vs / ys - gamma * vhy / ys;
        beta = -gamma * vs / ys;
    }
    for (int i = 0; i < n; i++) out[i] = gamma * hv[i] + delta * sj[i] + beta * hy[i];
}

/* Preconditioner solve, two-step self-scaled BFGS (ref: :1444-1528) */
static void t_msolve(tnc_ws *s, const real_t *g, real_t *y, bool upd1, real_t yksk, real_t yrsr,
                     bool lreset)
{
    const int n = s->n;
    if (upd1) { for (int i = 0; i < n; i++) y[i] = g[i] / s->diagb[i]; return; }
    real_t gsk = vdot(n, g, s->sk);
    real_t *hg = s->aux0, *hyr = s->aux1, *hyk = s->aux2;
    if (lreset) {
        for (int i = 0; i < n; i++) {
            real_t rd = 1.0 / s->diagb[i];
            hg[i] = g[i] * rd;
            hyk[i] = s->yk[i] * rd;
        }
        real_t ykhyk = vdot(n, s->yk, hyk);
        real_t ghyk = vdot(n, g, hyk);
        t_ssbfgs(n, s->sk, hg, hyk, yksk, ykhyk, gsk, ghyk, y);
    } else {
        for (int i = 0; i < n; i++) {
            real_t rd = 1.0 / s->diagb[i];
            hg[i] = g[i] * rd;
            hyk[i] = s->yk[i] * rd;
            hyr[i] = s->yr[i] * rd;
        }
        real_t gsr = vdot(n, g, s->sr);
        real_t ghyr = vdot(n, g, hyr);
        real_t yrhyr = vdot(n, s->yr, hyr);
        t_ssbfgs(n, s->sr, hg, hyr, yrsr, yrhyr, gsr, ghyr, hg);
        real_t yksr = vdot(n, s->yk, s->sr);
        real_t ykhyr = vdot(n, s->yk, hyr);
        t_ssbfgs(n, s->sr, hyk, hyr, yrsr, yrhyr, yksr, ykhyr, hyk);
        real_t ykhyk = vdot(n, hyk, s->yk);
        real_t ghyk = vdot(n, hyk, g);
        t_ssbfgs(n, s->sk, hg, hyk, yksk, ykhyk, gsk, ghyk, y);
    }
}

/* Diagonal preconditioner initialisation (ref: :1580-1658) */
static void t_init_precond(tnc_ws *s, bool lreset, real_t yksk, real_t yrsr, bool upd1)
{
    const int n = s->n;
    real_t *emat = s->emat, *diagb = s->diagb, *bsk = s->aux0;
    if (upd1) { memcpy(emat, diagb, sizeof(real_t) * (size_t)n); return; }
    if (lreset) {
        for (int i = 0; i < n; i++) bsk[i] = diagb[i] * s->sk[i];
        real_t sds = vdot(n, s->sk, bsk);
        if (yksk == 0.0) yksk = 1.0;
        if (sds == 0.0) sds = 1.0;
        for (int i = 0; i < n; i++) {
            real_t td = diagb[i];
            emat[i] = td - td * td * s->sk[i] * s->sk[i] / sds + s->yk[i] * s->yk[i] / yksk;
        }
    } else {
        for (int i = 0; i < n; i++) bsk[i] = diagb[i] * s->sr[i];
        real_t sds = vdot(n, s->sr, bsk);
        real_t srds = vdot(n, s->sk, bsk);
        real_t yrsk = vdot(n, s->yr, s->sk);
        if (yrsr == 0.0) yrsr = 1.0;
        if (sds == 0.0) sds = 1.0;
        for (int i = 0; i < n; i++) {
            real_t td = diagb[i];
            bsk[i] = td * s->sk[i] - bsk[i] * srds / sds + s->yr[i] * yrsk / yrsr;
            emat[i] = td - td * td * s->sr[i] * s->sr[i] / sds + s->yr[i] * s->yr[i] / yrsr;
        }
        sds = vdot(n, s->sk, bsk);
        if (yksk == 0.0) yksk = 1.0;
        if (sds == 0.0) sds = 1.0;
        for (int i = 0; i < n; i++)
            emat[i] -= bsk[i] * bsk[i] / sds + s->yk[i] * s->yk[i] / yksk;
    }
}

/* Forward-difference Hessian-vector product: one fused f/g evaluation (ref: :1388-1435) */
static void t_hess_vec(tnc_ws *s, const real_t *x, const real_t *g, real_t fscale, real_t accuracy,
                       real_t xnorm)
{
    const int n = s->n;
    real_t *xv = s->aux0, *gv = s->gv;
    const real_t *v = s->v;
    real_t delta = accuracy * (xnorm + 1.0);
    for (int i = 0; i < n; i++) xv[i] = x[i] + delta * v[i];
    t_unscale(n, xv, s->xscale, s->xoffset);
    t_clamp0(n, xv);
    real_t f;
    row_fun_grad(s->prob, xv, &f, gv);
    t_scaleg(n, gv, s->xscale, fscale);
    real_t dinv = 1.0 / delta;
    for (int i = 0; i < n; i++) gv[i] = (gv[i] - g[i]) * dinv;
}

/* Preconditioned linear CG on the Newton equations (ref: :1162-1341) */
static void t_direction(tnc_ws *s, real_t *zsol, const real_t *x, const real_t *g, int maxCGit,
                        bool upd1, real_t yksk, real_t yrsr, bool lreset, real_t fscale,
                        real_t accuracy, real_t gnorm, real_t xnorm)
{
    const int n = s->n;
    const int *pivot = s->pivot;
    real_t *r = s->r, *v = s->v, *zk = s->zk, *gv = s->gv, *emat = s->emat;

    if (maxCGit == 0) {
        for (int i = 0; i < n; i++) zsol[i] = -g[i];
        t_project(n, zsol, pivot);
        return;
    }
    real_t rhsnrm = gnorm, tol = 1e-12, qold = 0.0, rzold = 0.0;

    t_init_precond(s, lreset, yksk, yrsr, upd1);
    for (int i = 0; i < n; i++) { r[i] = -g[i]; v[i] = 0.0; zsol[i] = 0.0; }

    for (int it = 0; it < maxCGit; it++) {
        t_project(n, r, pivot);
        t_msolve(s, r, zk, upd1, yksk, yrsr, lreset);
        t_project(n, zk, pivot);
        real_t rz = vdot(n, r, zk);

        if ((rz / rhsnrm < tol) || (s->nfeval >= (s->maxnfeval - 1))) {
            if (it == 0) {
                for (int i = 0; i < n; i++) zsol[i] = -g[i];
                t_project(n, zsol, pivot);
            }
            break;
        }
        real_t beta = (it == 0) ? (real_t)0.0 : rz / rzold;
        for (int i = 0; i < n; i++) v[i] = zk[i] + beta * v[i];
        t_project(n, v, pivot);

        t_hess_vec(s, x, g, fscale, accuracy, xnorm);
        s->nfeval++;
        t_project(n, gv, pivot);

        real_t vgv = vdot(n, v, gv);
        if (vgv / rhsnrm < tol) {
            if (it == 0) {
                t_msolve(s, g, zsol, upd1, yksk, yrsr, lreset);
                for (int i = 0; i < n; i++) zsol[i] = -zsol[i];
                t_project(n, zsol, pivot);
            }
            break;
        }
        /* diagonal BFGS-like scaling update (ref: :1347-1362) */
        {
            real_t vr = 1.0 / vdot(n, v, r);
            real_t ivgv = 1.0 / vdot(n, v, gv);
            for (int i = 0; i < n; i++) {
                emat[i] += -r[i] * r[i] * vr + gv[i] * gv[i] * ivgv;
                emat[i] = (emat[i] <= 1e-6) ? (real_t)1. : emat[i];
            }
        }
        real_t alpha = rz / vgv;
        vaxpy(n, alpha, v, zsol);
        vaxpy(n, -alpha, gv, r);

        real_t gtp = vdot(n, zsol, g);
        real_t pr = vdot(n, r, zsol);
        real_t qnew = (gtp + pr) * 0.5;
        real_t qtest = (it + 1) * (1.0 - qold / qnew);
        if (qtest <= 0.5) break;
        if (gtp > 0.0) { vaxpy(n, -alpha, v, zsol); break; }
        qold = qnew;
        rzold = rz;
    }
    memcpy(s->diagb, emat, sizeof(real_t) * (size_t)n);
}

/* Safeguarded cubic step-length finder of Gill & Murray, kept as one state record instead of the
   reference's 17 by-pointer scalars (ref: getptcInit :1822-1888, getptcIter :1890-2154). */
typedef struct {
    real_t reltol, abstol, tnytol, fpresn, xbnd, big, rtsmll;
    real_t u, fu, gu, xmin, fmin, gmin, xw, fw, gw, a, b, oldf, b1, scxbnd, e, step, factor;
    real_t gtest1, gtest2, tol;
    bool braktd;
} ptc_t;
enum { PTC_OK = 0, PTC_EVAL = 1, PTC_EINVAL = 2, PTC_FAIL = 3 };

static void ptc_clip_step(ptc_t *q) /* shared tail of init and iter (ref: :1875-1886, :2141-2152) */
{
    if (q->step >= q->scxbnd) {
        q->step = q->scxbnd;
        q->scxbnd -= (q->reltol * fabs(q->xbnd) + q->abstol) / (1.0 + q->reltol);
    }
    q->u = q->step;
    if (fabs(q->step) < q->tol && q->step < 0.0) q->u = -q->tol;
    if (fabs(q->step) < q->tol && q->step >= 0.0) q->u = q->tol;
}

static int ptc_init(ptc_t *q, real_t eta, real_t rmu)
{
    if (q->u <= 0.0 || q->xbnd <= q->tnytol || q->gu > 0.0) return PTC_EINVAL;
    if (q->xbnd < q->abstol) q->abstol = q->xbnd;
    q->tol = q->abstol;
    q->a = 0.0; q->xw = 0.0; q->xmin = 0.0;
    q->oldf = q->fu; q->fmin = q->fu; q->fw = q->fu;
    q->gw = q->gu; q->gmin = q->gu;
    q->step = q->u;
    q->factor = 5.0;
    q->braktd = false;
    q->scxbnd = q->xbnd;
    q->b = q->scxbnd + q->reltol * fabs(q->scxbnd) + q->abstol;
    q->e = q->b + q->b;
    q->b1 = q->b;
    q->gtest1 = -rmu * q->gu;
    q->gtest2 = -eta * q->gu;
    ptc_clip_step(q);
    return PTC_EVAL;
}

static int ptc_iter(ptc_t *q)
{
    real_t r = 0.0, qq = 0.0, s = 0.0, a1, xmidpt, twotol;
    bool skip_update = false;

    if (q->fu <= q->fmin) {
        real_t chordu = q->oldf - (q->xmin + q->u) * q->gtest1;
        if (q->fu > chordu) {
            /* not a sufficient decrease: fabricate (fu, gu) so the interpolation bisects or takes
               the chord root (ref: :1910-1932) */
            real_t chordm = q->oldf - q->xmin * q->gtest1;
            q->gu = -q->gmin;
            real_t denom = chordm - q->fmin;
            if (fabs(denom) < 1e-15) {
                denom = 1e-15;
                if (chordm - q->fmin < 0.0) denom = -denom;
            }
            if (q->xmin != 0.0) q->gu = q->gmin * (chordu - q->fu) / denom;
            q->fu = 0.5 * q->u * (q->gmin + q->gu) + q->fmin;
            if (q->fu < q->fmin) q->fu = q->fmin;
        } else {
            /* new lowest point becomes the origin (ref: :1933-1952) */
            q->fw = q->fmin; q->fmin = q->fu;
            q->gw = q->gmin; q->gmin = q->gu;
            q->xmin += q->u;
            q->a -= q->u; q->b -= q->u;
            q->xw = -q->u;
            q->scxbnd -= q->u;
            if (q->gu <= 0.0) q->a = 0.0;
            else { q->b = 0.0; q->braktd = true; }
            q->tol = fabs(q->xmin) * q->reltol + q->abstol;
            skip_update = true;
        }
    }
    if (!skip_update) { /* origin unchanged, new point may become w (ref: :1957-1966) */
        if (q->u < 0.0) q->a = q->u;
        else { q->b = q->u; q->braktd = true; }
        q->xw = q->u; q->fw = q->fu; q->gw = q->gu;
    }

    twotol = q->tol + q->tol;
    xmidpt = 0.5 * (q->a + q->b);

    bool convrg = (fabs(xmidpt) <= twotol - 0.5 * (q->b - q->a)) ||
                  (fabs(q->gmin) <= q->gtest2 && q->fmin < q->oldf &&
                   ((fabs(q->xmin - q->xbnd) > q->tol) || (!q->braktd)));
    if (convrg) {
        if (q->xmin != 0.0) return PTC_OK;
        if (fabs(q->oldf - q->fw) <= q->fpresn) return PTC_FAIL;
        q->tol = 0.1 * q->tol;
        if (q->tol < q->tnytol) return PTC_FAIL;
        q->reltol = 0.1 * q->reltol;
        q->abstol = 0.1 * q->abstol;
        twotol = 0.1 * twotol;
    }

    bool minimum_found = false;
    if (fabs(q->e) > q->tol) {
        /* cubic through xmin and xw (ref: :2003-2075) */
        r = 3.0 * (q->fmin - q->fw) / q->xw + q->gmin + q->gw;
        real_t absr = fabs(r);
        qq = absr;
        if (q->gw != 0.0 && q->gmin != 0.0) {
            real_t abgw = fabs(q->gw), abgmin = fabs(q->gmin);
            s = sqrt(abgmin) * sqrt(abgw);
            if (q->gw / abgw * q->gmin > 0.0) {
                if (r >= s || r <= -s) {
                    qq = sqrt(fabs(r + s)) * sqrt(fabs(r - s));
                } else {
                    r = 0.0; qq = 0.0;
                    minimum_found = true;
                }
            } else {
                real_t sumsq = 1.0, pp = 0.0, scale;
                if (absr >= s) {
                    if (absr > q->rtsmll) pp = absr * q->rtsmll;
                    if (s >= pp) { real_t val = s / absr; sumsq = 1.0 + val * val; }
                    scale = absr;
                } else {
                    if (s > q->rtsmll) pp = s * q->rtsmll;
                    if (absr >= pp) { real_t val = absr / s; sumsq = 1.0 + val * val; }
                    scale = s;
                }
                sumsq = sqrt(sumsq);
                qq = q->big;
                if (scale < q->big / sumsq) qq = scale * sumsq;
            }
        }
        if (!minimum_found) {
            if (q->xw < 0.0) qq = -qq;
            s = q->xw * (q->gmin - r - qq);
            qq = q->gw - q->gmin + qq + qq;
            if (qq > 0.0) s = -s;
            if (qq <= 0.0) qq = -qq;
            r = q->e;
            if (q->b1 != q->step || q->braktd) q->e = q->step;
        }
    }

    /* artificial bound on the step (ref: :2077-2114) */
    a1 = q->a;
    q->b1 = q->b;
    q->step = xmidpt;
    if ((!q->braktd) || ((q->a == 0.0 && q->xw < 0.0) || (q->b == 0.0 && q->xw > 0.0))) {
        if (q->braktd) {
            real_t d1 = q->xw, d2 = q->a;
            if (q->a == 0.0) d2 = q->b;
            q->u = -d1 / d2;
            q->step = 5.0 * d2 * (0.1 + 1.0 / q->u) / 11.0;
            if (q->u < 1.0) q->step = 0.5 * d2 * sqrt(q->u);
        } else {
            q->step = -q->factor * q->xw;
            if (q->step > q->scxbnd) q->step = q->scxbnd;
            if (q->step != q->scxbnd) q->factor = 5.0 * q->factor;
        }
        if (q->step <= 0.0) a1 = q->step;
        if (q->step > 0.0) q->b1 = q->step;
    }

    /* accept or reject the interpolated step (ref: :2121-2137) */
    if (fabs(s) <= fabs(0.5 * qq * r) || s <= qq * a1 || s >= qq * q->b1) {
        q->e = q->b - q->a;
    } else {
        q->step = s / qq;
        if (q->step - q->a < twotol || q->b - q->step < twotol) {
            if (xmidpt <= 0.0) q->step = -q->tol;
            else q->step = q->tol;
        }
    }
    ptc_clip_step(q);
    return PTC_EVAL;
}

enum { LS_OK = 0, LS_MAXFUN = 1, LS_FAIL = 2 };

/* Line search along p (ref: linearSearch :1664-1813; maxlsit = 64 at :1676).  gfull is the
   unscaled gradient at x and is replaced by the gradient at the accepted point. */
static int t_linesearch(tnc_ws *s, real_t fscale, real_t eta, real_t ftol, real_t xbnd,
                        const real_t *p, real_t *x, real_t *f, real_t *alpha, real_t *gfull)
{
    const int n = s->n;
    real_t *temp = s->r, *tempg = s->v, *newg = s->zk; /* ref: :1700-1703 reuses this region */
    const int maxlsit = 64;
    ptc_t q;

    memcpy(temp, gfull, sizeof(real_t) * (size_t)n);
    t_scaleg(n, temp, s->xscale, fscale);
    q.gu = vdot(n, temp, p);

    memcpy(temp, x, sizeof(real_t) * (size_t)n);
    t_project(n, temp, s->pivot);
    real_t xnorm = vnrm2(n, temp);

    real_t rteps = sqrt(R_EPS);
    real_t pe = vnrm2(n, p) + R_EPS;
    q.reltol = rteps * (xnorm + 1.0) / pe;
    q.abstol = -R_EPS * (1.0 + fabs(*f)) / (q.gu - R_EPS);
    q.tnytol = R_EPS * (xnorm + 1.0) / pe;
    q.rtsmll = R_EPS;
    q.big = 1.0 / (R_EPS * R_EPS);
    q.fpresn = ftol;
    q.xbnd = xbnd;
    q.u = *alpha;
    q.xmin = *alpha; /* the reference passes `alpha` itself as xmin (ref: :1738) */
    q.fu = *f;
    q.fmin = *f;
    const real_t rmu = 1e-4;

    int itcnt = 0;
    int itest = ptc_init(&q, eta, rmu);

    while (itest == PTC_EVAL) {
        if ((++itcnt > maxlsit) || (s->nfeval >= s->maxnfeval)) break;
        real_t ualpha = q.xmin + q.u;
        for (int i = 0; i < n; i++) temp[i] = x[i] + ualpha * p[i];
        t_unscale(n, temp, s->xscale, s->xoffset);
        t_clamp0(n, temp);
        row_fun_grad(s->prob, temp, &q.fu, tempg);
        s->nfeval++;
        q.fu *= fscale;
        memcpy(temp, tempg, sizeof(real_t) * (size_t)n);
        t_scaleg(n, temp, s->xscale, fscale);
        q.gu = vdot(n, temp, p);
        itest = ptc_iter(&q);
        if (q.xmin == ualpha) memcpy(newg, tempg, sizeof(real_t) * (size_t)n);
    }
    *alpha = q.xmin;

    if (itest == PTC_OK) {
        *f = q.fmin;
        vaxpy(n, *alpha, p, x);
        memcpy(gfull, newg, sizeof(real_t) * (size_t)n);
        return LS_OK;
    }
    if (itcnt > maxlsit) return LS_FAIL;
    if (itest != PTC_EVAL) return LS_FAIL;
    return LS_MAXFUN;
}

/* One row through TNC with tncg_iteration's constants (ref: src/poismf.c:383-391):
   eta .25, stepmx 10, accuracy 0 -> sqrt(eps), fmin 0, ftol 1e-4, xtol -1 -> sqrt(eps),
   pgtol -1 -> 1e-2 sqrt(accuracy), rescale 1.3.  buffer = 19k reals (+ gfull k), pivot = k ints. */
static int row_tnc(const rowprob *prob, real_t *x, int maxCGit, int maxnfeval, real_t *buffer,
                   int *pivot, real_t *f_out, int *nfeval_out, int *niter_out)
{
    const int n = prob->k;
    tnc_ws S;
    tnc_ws *s = &S;
    real_t *b = buffer;
    s->prob = prob; s->n = n; s->pivot = pivot;
    s->xscale = b; b += n; s->xoffset = b; b += n;
    s->oldg = b; b += n; s->g = b; b += n; s->temp = b; b += n; s->diagb = b; b += n;
    s->pk = b; b += n; s->sk = b; b += n; s->yk = b; b += n; s->sr = b; b += n; s->yr = b; b += n;
    s->r = b; b += n; s->v = b; b += n; s->zk = b; b += n; s->emat = b; b += n; s->gv = b; b += n;
    s->aux0 = b; b += n; s->aux1 = b; b += n; s->aux2 = b; b += n;
    real_t *gfull = b; /* 20th vector */
    s->nfeval = 0; s->maxnfeval = maxnfeval;

    real_t f;
    *niter_out = 0;
    t_clamp0(n, x);                              /* ref: src/tnc.c:323 */
    if (maxnfeval < 1) { *nfeval_out = 0; *f_out = 0; return T_MAXFUN; }
    row_fun_grad(prob, x, &f, gfull);            /* ref: :341 */
    s->nfeval++;

    for (int i = 0; i < n; i++) {                /* ref: :383-399 (Q9) */
        s->xscale[i] = 1.0 + fabs(x[i]);
        s->xoffset[i] = x[i];
    }
    real_t fscale = 1.0;
    real_t rteps = sqrt(R_EPS);                  /* ref: :402-436 with poismf's arguments */
    real_t stepmx = 10., eta = 0.25, rescale = 1.3, accuracy = 0., fmin_est = 0., ftol = 1e-4,
           xtol = -1., pgtol = -1.;
    if (stepmx < rteps * 10.0) stepmx = 1.0e1;
    if (maxCGit > n) maxCGit = n;
    if (accuracy <= R_EPS) accuracy = rteps;
    if (pgtol < 0.0) pgtol = 1e-2 * sqrt(accuracy);
    if (xtol < 0.0) xtol = rteps;

    /* ---- tnc_minimize (ref: :554-993) ---- */
    real_t *g = s->g, *oldg = s->oldg, *temp = s->temp, *diagb = s->diagb, *pk = s->pk;
    real_t *sk = s->sk, *yk = s->yk, *sr = s->sr, *yr = s->yr;
    real_t difnew = 0.0, epsred = 0.05, difold, oldf, oldgtp, xnorm, gnorm, ustpmax, spe;
    real_t fLastReset, fLastConstraint, yrsr = 0.0, yksk = 0.0, alpha = 0.0;
    bool upd1 = true, newcon = true, lreset = false, remcon;
    int icycle = n - 1, niter = 0, rc;

    for (int i = 0; i < n; i++)                  /* scalex, ref: :492-501 */
        if (s->xscale[i] > 0.0) x[i] = (x[i] - s->xoffset[i]) / s->xscale[i];
    f *= fscale;

    for (int i = 0; i < n; i++) {                /* setConstraints with low = 0, ref: :513-545 */
        if (s->xscale[i] == 0.0) pivot[i] = 2;
        else if (x[i] * s->xscale[i] + s->xoffset[i] - (real_t)0. <= R_EPS * 10.0 * (fabs((real_t)0.) + 1.0))
            pivot[i] = -1;
        else pivot[i] = 0;
    }
    memcpy(g, gfull, sizeof(real_t) * (size_t)n);
    t_scaleg(n, g, s->xscale, fscale);
    for (int i = 0; i < n; i++) if (-pivot[i] * g[i] < 0.0) pivot[i] = 0; /* ref: :670-674 */
    t_project(n, g, pivot);
    gnorm = vnrm2(n, g);
    fLastConstraint = f;
    fLastReset = f;
    for (int i = 0; i < n; i++) diagb[i] = 1.0;

    for (;;) {
        if (vnrm2(n, g) <= pgtol * fscale) {     /* ref: :700-712 */
            memcpy(g, gfull, sizeof(real_t) * (size_t)n);
            t_project(n, g, pivot);
            rc = T_LOCALMINIMUM;
            break;
        }
        if (s->nfeval >= maxnfeval) { rc = T_MAXFUN; break; }

        real_t newscale = vnrm2(n, g);           /* ref: :720-746 */
        if ((newscale > R_EPS) && (fabs(log10(newscale)) > rescale)) {
            newscale = 1.0 / newscale;
            f *= newscale; fscale *= newscale; gnorm *= newscale;
            fLastConstraint *= newscale; fLastReset *= newscale; difnew *= newscale;
            for (int i = 0; i < n; i++) g[i] *= newscale;
            for (int i = 0; i < n; i++) diagb[i] = 1.0;
            upd1 = true; icycle = n - 1; newcon = true;
        }

        memcpy(temp, x, sizeof(real_t) * (size_t)n);
        t_project(n, temp, pivot);
        xnorm = vnrm2(n, temp);
        int oldnfeval = s->nfeval;

        t_direction(s, pk, x, g, maxCGit, upd1, yksk, yrsr, lreset, fscale, accuracy, gnorm, xnorm);

        if (!newcon) {                           /* ref: :770-785 */
            if (!lreset) {
                vaxpy(n, (real_t)1., sk, sr);
                vaxpy(n, (real_t)1., yk, yr);
                icycle++;
            } else {
                memcpy(sr, sk, sizeof(real_t) * (size_t)n);
                memcpy(yr, yk, sizeof(real_t) * (size_t)n);
                fLastReset = f;
                icycle = 1;
            }
        }
        memcpy(oldg, g, sizeof(real_t) * (size_t)n);
        oldf = f;
        oldgtp = vdot(n, pk, g);

        ustpmax = stepmx / (vnrm2(n, pk) + R_EPS);
        spe = ustpmax;                           /* stepMax with low = 0, up = +inf, ref: :1041-1067 */
        for (int i = 0; i < n; i++) {
            if ((pivot[i] == 0) && (pk[i] != 0.0) && (pk[i] < 0.0)) {
                real_t t = ((real_t)0. - s->xoffset[i]) / s->xscale[i] - x[i];
                if (t > spe * pk[i]) spe = t / pk[i];
            }
        }

        if (spe > 0.0) {
            /* initialStep, ref: :1368-1383 */
            {
                real_t d = fabs(f - fmin_est / fscale);
                alpha = 1.0;
                if (d * 2.0 <= -oldgtp && d >= R_EPS) alpha = d * -2.0 / oldgtp;
                if (alpha >= spe) alpha = spe;
            }
            int lsrc = t_linesearch(s, fscale, eta, ftol, spe, pk, x, &f, &alpha, gfull);
            if (lsrc == LS_FAIL) { rc = T_LSFAIL; break; }
            if (alpha >= 0.9 * ustpmax) stepmx *= 1e2;
            if (alpha - spe >= -R_EPS * 10.0) newcon = true;
            else {
                if (lsrc != LS_OK) { rc = (lsrc == LS_MAXFUN) ? T_MAXFUN : T_LSFAIL; break; }
                newcon = false;
            }
        } else {
            newcon = true;
        }

        if (newcon) {                            /* addConstraint with low = 0, ref: :1072-1108 */
            bool added = false;
            for (int i = 0; i < n; i++) {
                if ((pivot[i] == 0) && (pk[i] != 0.0) && (pk[i] < 0.0)) {
                    real_t tolc = R_EPS * 10.0 * (fabs((real_t)0.) + 1.0);
                    if (x[i] * s->xscale[i] + s->xoffset[i] - (real_t)0. <= tolc) {
                        pivot[i] = -1;
                        x[i] = ((real_t)0. - s->xoffset[i]) / s->xscale[i];
                        added = true;
                    }
                }
            }
            if (!added && s->nfeval == oldnfeval) { rc = T_NOPROGRESS; break; }
            fLastConstraint = f;
        }
        niter++;

        difold = difnew;
        difnew = oldf - f;
        if (icycle == 1) {
            if (difnew > difold * 2.0) epsred += epsred;
            if (difnew < difold * 0.5) epsred *= 0.5;
        }

        memcpy(g, gfull, sizeof(real_t) * (size_t)n);
        t_scaleg(n, g, s->xscale, fscale);
        memcpy(temp, g, sizeof(real_t) * (size_t)n);
        t_project(n, temp, pivot);
        gnorm = vnrm2(n, temp);

        /* removeConstraint, ref: :1113-1153 */
        remcon = false;
        if (!(((fLastConstraint - f) <= (oldgtp * -0.5)) && (gnorm > pgtol * fscale))) {
            int imax = -1;
            real_t cmax = 0.0;
            for (int i = 0; i < n; i++) {
                if (pivot[i] == 2) continue;
                real_t t = -pivot[i] * g[i];
                if (t < cmax) { cmax = t; imax = i; }
            }
            if (imax != -1) { pivot[imax] = 0; remcon = true; }
        }
        if (remcon) {
            memcpy(temp, g, sizeof(real_t) * (size_t)n);
            t_project(n, temp, pivot);
            gnorm = vnrm2(n, temp);
            fLastConstraint = f;
        }

        if (!remcon && !newcon) {                /* ref: :909-929 */
            if (fabs(difnew) <= ftol * fscale) { rc = T_FCONVERGED; break; }
            if (alpha * vnrm2(n, pk) <= xtol) { rc = T_XCONVERGED; break; }
        }
        t_project(n, g, pivot);

        if (!newcon) {                           /* ref: :940-962 */
            for (int i = 0; i < n; i++) {
                yk[i] = g[i] - oldg[i];
                sk[i] = alpha * pk[i];
            }
            yksk = vdot(n, yk, sk);
            if (icycle == (n - 1) || difnew < epsred * (fLastReset - f)) lreset = true;
            else {
                yrsr = vdot(n, yr, sr);
                lreset = (yrsr <= 0.0);
            }
            upd1 = false;
        }
    }

    t_unscale(n, x, s->xscale, s->xoffset);      /* ref: :971-973 */
    t_clamp0(n, x);
    f /= fscale;
    *f_out = f;
    *nfeval_out = s->nfeval;
    *niter_out = niter;
    return rc;
}

static int tnc_max_cg_iters(size_t k) /* ref: src/poismf.c:342 */
{
    return (int)fmax(1., fmin(50., (real_t)k / 2.));
}

/* G2 entry point: one row through TNC */
ORC_API int oracle_tnc_row(real_t *x, const real_t *F, const real_t *bsum, const real_t *xval,
                           const sparse_ix *xind, size_t nnz, int k, real_t l2, real_t w, int maxupd,
                           real_t *f_out, int *nfeval_out, int *niter_out)
{
    rowprob p = mk_prob(F, bsum, xval, xind, nnz, k, l2, w);
    real_t *buffer = (real_t *)malloc(sizeof(real_t) * 22 * (size_t)k);
    int *pivot = (int *)malloc(sizeof(int) * (size_t)k);
    int rc = row_tnc(&p, x, tnc_max_cg_iters((size_t)k), maxupd, buffer, pivot, f_out, nfeval_out,
                     niter_out);
    free(buffer);
    free(pivot);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Half-sweep drivers: every row of M independently against the fixed factor F                */
/* ------------------------------------------------------------------------------------------ */

/* ref: src/poismf.c:77-83 (sum_by_cols): serial, in row order */
ORC_API void oracle_sum_by_cols(real_t *out, const real_t *M, size_t nrow, size_t ncol)
{
    for (size_t c = 0; c < ncol; c++) out[c] = 0;
    for (size_t r = 0; r < nrow; r++)
        for (size_t c = 0; c < ncol; c++) out[c] += M[r * ncol + c];
}

/* ref: src/poismf.c:85-123 (adjustment_Bsum): Bsum_w[r] = (w-1) sum_{j in nz(r)} F_j + Bsum */
ORC_API void oracle_adjustment_bsum(const real_t *F, const real_t *bsum, real_t *bsum_w,
                                    const sparse_ix *indices, const sparse_ix *indptr, size_t dimM,
                                    size_t k, real_t w, int nthreads)
{
    real_t wm1 = w - 1.;
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (size_t r = 0; r < dimM; r++) {
        real_t *o = bsum_w + r * k;
        for (size_t c = 0; c < k; c++) o[c] = 0;
        for (size_t j = indptr[r]; j < indptr[r + 1]; j++)
            vaxpy((int)k, (real_t)1., F + (size_t)indices[j] * k, o);
        for (size_t c = 0; c < k; c++) o[c] *= wm1;
        vaxpy((int)k, (real_t)1., bsum, o);
    }
}

/* ref: src/poismf.c:139-188 (pg_iteration).  `cnst_sum` arrives pre-scaled by the caller. */
ORC_API void oracle_pg_iteration(real_t *M, const real_t *F, const real_t *xval,
                                 const sparse_ix *indptr, const sparse_ix *indices, size_t dimM,
                                 size_t k, real_t cnst_div, const real_t *cnst_sum,
                                 const real_t *bsum_w, real_t step, real_t w, size_t maxupd,
                                 int nthreads)
{
    step *= w; /* ref: :151 */
    #pragma omp parallel num_threads(nthreads)
    {
        real_t *grad = (real_t *)malloc(sizeof(real_t) * k);
        #pragma omp for schedule(dynamic)
        for (size_t r = 0; r < dimM; r++) {
            real_t *a = M + r * k;
            size_t nnz = indptr[r + 1] - indptr[r];
            if (nnz == 0) { memset(a, 0, sizeof(real_t) * k); continue; } /* Q7 */
            const real_t *shift = (w != 1.) ? bsum_w + r * k : cnst_sum;
            for (size_t u = 0; u < maxupd; u++) {
                oracle_calc_grad_pgd(grad, a, F, xval + indptr[r], indices + indptr[r], nnz, (int)k);
                vaxpy((int)k, step, grad, a);
                vaxpy((int)k, (real_t)1., shift, a);
                vscal((int)k, cnst_div, a);
                for (size_t c = 0; c < k; c++) a[c] = (a[c] > 0.) ? a[c] : (real_t)0.;
            }
        }
        free(grad);
    }
}

/* ref: src/poismf.c:275-322 (cg_iteration) */
ORC_API void oracle_cg_iteration(real_t *M, const real_t *F, const real_t *xval,
                                 const sparse_ix *indptr, const sparse_ix *indices, size_t dimM,
                                 size_t k, int limit_step, const real_t *bsum, real_t l2, real_t w,
                                 size_t maxupd, const real_t *bsum_w, int nthreads)
{
    #pragma omp parallel num_threads(nthreads)
    {
        real_t *scratch = (real_t *)malloc(sizeof(real_t) * 5 * k);
        #pragma omp for schedule(dynamic)
        for (size_t r = 0; r < dimM; r++) {
            real_t *a = M + r * k;
            size_t nnz = indptr[r + 1] - indptr[r];
            if (nnz == 0) { memset(a, 0, sizeof(real_t) * k); continue; }
            rowprob p = mk_prob(F, (w != 1.) ? bsum_w + r * k : bsum, xval + indptr[r],
                                indices + indptr[r], nnz, (int)k, l2, w);
            real_t fv; size_t ni, nf;
            row_cg(&p, a, w != 1., (real_t)1e-2, 150, maxupd, (real_t)0.25, (real_t)0.01, 20,
                   limit_step != 0, scratch, &fv, &ni, &nf);
        }
        free(scratch);
    }
}

/* ref: src/poismf.c:324-404 (tncg_iteration).  Returns has_converged. */
ORC_API int oracle_tncg_iteration(real_t *M, const real_t *F, int reuse_prev, const real_t *xval,
                                  const sparse_ix *indptr, const sparse_ix *indices, size_t dimM,
                                  size_t k, const real_t *bsum, real_t l2, real_t w, int maxupd,
                                  int early_stop, const real_t *bsum_w, int nthreads)
{
    size_t n_unchanged = 0;
    int maxCGit = tnc_max_cg_iters(k);
    #pragma omp parallel num_threads(nthreads) reduction(+:n_unchanged)
    {
        real_t *buffer = (real_t *)malloc(sizeof(real_t) * 23 * k);
        int *pivot = (int *)malloc(sizeof(int) * k);
        real_t *prev = buffer + 22 * k;
        #pragma omp for schedule(dynamic)
        for (size_t r = 0; r < dimM; r++) {
            real_t *a = M + r * k;
            size_t nnz = indptr[r + 1] - indptr[r];
            if (nnz == 0) { memset(a, 0, sizeof(real_t) * k); continue; }
            rowprob p = mk_prob(F, (w != 1.) ? bsum_w + r * k : bsum, xval + indptr[r],
                                indices + indptr[r], nnz, (int)k, l2, w);
            if (early_stop) memcpy(prev, a, sizeof(real_t) * k);
            if (!reuse_prev) for (size_t c = 0; c < k; c++) a[c] = 1e-3;
            real_t fv; int nf, ni;
            row_tnc(&p, a, maxCGit, maxupd, buffer, pivot, &fv, &nf, &ni);
            if (early_stop) {
                vaxpy((int)k, (real_t)-1., a, prev);
                n_unchanged += vdot((int)k, prev, prev) <= 1e-4;
            }
        }
        free(buffer);
        free(pivot);
    }
    if (early_stop) return ((double)n_unchanged / (double)dimM) >= .95;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Outer alternation                                    ref: src/poismf.c:435-632 (run_poismf) */
/* ------------------------------------------------------------------------------------------ */
enum { M_TNCG = 1, M_CG = 2, M_PG = 3 }; /* ref: src/poismf.h:225 */

ORC_API int oracle_run_poismf(real_t *A, real_t *Xr, sparse_ix *Xr_indptr, sparse_ix *Xr_indices,
                              real_t *B, real_t *Xc, sparse_ix *Xc_indptr, sparse_ix *Xc_indices,
                              const size_t dimA, const size_t dimB, const size_t k,
                              const real_t l2_reg, const real_t l1_reg, const real_t w_mult,
                              real_t step_size, const int method, const bool limit_step,
                              const size_t numiter, const size_t maxupd, const bool early_stop,
                              const bool reuse_prev, const bool handle_interrupt, const int nthreads)
{
    (void)handle_interrupt;
    real_t *cnst_sum = (real_t *)malloc(sizeof(real_t) * k);
    real_t *bsum_w = NULL;
    if (w_mult != 1.) bsum_w = (real_t *)malloc(sizeof(real_t) * k * (dimA > dimB ? dimA : dimB));
    if (cnst_sum == NULL || (w_mult != 1. && bsum_w == NULL)) { free(cnst_sum); free(bsum_w); return 1; }
    real_t neg_step = -step_size;
    bool stopA = false, stopB = false;

    for (size_t it = 0; it < numiter; it++) {
        /* Q6: cnst_div uses the step before halving and is reused by the A half */
        real_t cnst_div = 1. / (1. + 2. * l2_reg * step_size);

        /* ---- B half first (Q5) ---- */
        oracle_sum_by_cols(cnst_sum, A, dimA, k);
        if (l1_reg > 0.) for (size_t c = 0; c < k; c++) cnst_sum[c] += l1_reg;
        if (w_mult != 1.)
            oracle_adjustment_bsum(A, cnst_sum, bsum_w, Xc_indices, Xc_indptr, dimB, k, w_mult, nthreads);
        if (method == M_PG) {
            if (w_mult == 1.) vscal((int)k, neg_step, cnst_sum);
            else for (size_t i = 0; i < dimB * k; i++) bsum_w[i] *= neg_step;
            oracle_pg_iteration(B, A, Xc, Xc_indptr, Xc_indices, dimB, k, cnst_div, cnst_sum, bsum_w,
                                step_size, w_mult, maxupd, nthreads);
            step_size *= 0.5;
            neg_step = -step_size;
        } else if (method == M_CG) {
            oracle_cg_iteration(B, A, Xc, Xc_indptr, Xc_indices, dimB, k, limit_step, cnst_sum, l2_reg,
                                w_mult, maxupd, bsum_w, nthreads);
        } else {
            if (!stopB)
                stopB = oracle_tncg_iteration(B, A, reuse_prev, Xc, Xc_indptr, Xc_indices, dimB, k,
                                              cnst_sum, l2_reg, w_mult, (int)maxupd, early_stop,
                                              bsum_w, nthreads);
        }

        /* ---- A half ---- */
        oracle_sum_by_cols(cnst_sum, B, dimB, k);
        if (l1_reg > 0.) for (size_t c = 0; c < k; c++) cnst_sum[c] += l1_reg;
        if (w_mult != 1.)
            oracle_adjustment_bsum(B, cnst_sum, bsum_w, Xr_indices, Xr_indptr, dimA, k, w_mult, nthreads);
        if (method == M_PG) {
            if (w_mult == 1.) vscal((int)k, neg_step, cnst_sum);
            else for (size_t i = 0; i < dimA * k; i++) bsum_w[i] *= neg_step;
            vscal((int)k, neg_step, cnst_sum); /* Q1: second scaling, ref: :577 */
            oracle_pg_iteration(A, B, Xr, Xr_indptr, Xr_indices, dimA, k, cnst_div, cnst_sum, bsum_w,
                                step_size, w_mult, maxupd, nthreads);
        } else if (method == M_CG) {
            oracle_cg_iteration(A, B, Xr, Xr_indptr, Xr_indices, dimA, k, limit_step, cnst_sum, l2_reg,
                                w_mult, maxupd, bsum_w, nthreads);
        } else {
            if (!stopA)
                stopA = oracle_tncg_iteration(A, B, reuse_prev, Xr, Xr_indptr, Xr_indices, dimA, k,
                                              cnst_sum, l2_reg, w_mult, (int)maxupd, early_stop,
                                              bsum_w, nthreads);
        }
        if (stopA && stopB) break;
    }
    free(cnst_sum);
    free(bsum_w);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Factors for new rows with B fixed                    ref: src/pred.c:66-199 (factors_multiple) */
/* ------------------------------------------------------------------------------------------ */
/* `Bsum` arrives with l1 already added (ref: :80).  Quirks restated: (i) rows start at Amean unless the
   solver is TNCG without reuse_mean, which starts them at 1e-3 inside tncg_iteration (ref: :144-147);
   (ii) PG rescales Bsum by -step on every inner iteration and halves the step (ref: :152-167); with
   w_mult != 1 the per-row Bsum_w was already scaled once by -step at set-up (ref: :121-122), so it ends up
   scaled twice; (iii) CG runs ONE cg_iteration with maxiter = maxupd * niter (ref: :175-178). */
ORC_API int oracle_factors_multiple(real_t *A, real_t *B, real_t *Bsum, real_t *Amean, real_t *Xr,
                                    sparse_ix *Xr_indptr, sparse_ix *Xr_indices, int k, size_t dimA,
                                    real_t l2_reg, real_t w_mult, real_t step_size, size_t niter,
                                    size_t maxupd, int method, bool limit_step, bool reuse_mean, int nthreads)
{
    const size_t ks = (size_t)k;
    real_t *bsum_w = NULL, *bsum_w_scaled = NULL, *bsum_scaled = NULL;
    if (method == M_PG) {
        if (w_mult == 1.) bsum_scaled = (real_t *)malloc(sizeof(real_t) * ks);
        else bsum_w_scaled = (real_t *)malloc(sizeof(real_t) * ks * dimA);
    }
    if (w_mult != 1.) {
        bsum_w = (real_t *)malloc(sizeof(real_t) * ks * dimA);
        oracle_adjustment_bsum(B, Bsum, bsum_w, Xr_indices, Xr_indptr, dimA, ks, w_mult, nthreads);
        if (method == M_PG) for (size_t i = 0; i < dimA * ks; i++) bsum_w[i] *= -step_size;
    }
    if (reuse_mean || method != M_TNCG)
        for (size_t r = 0; r < dimA; r++) memcpy(A + r * ks, Amean, sizeof(real_t) * ks);

    if (method == M_PG) {
        for (size_t it = 0; it < niter; it++) {
            if (w_mult == 1.) {
                memcpy(bsum_scaled, Bsum, sizeof(real_t) * ks);
                vscal(k, -step_size, bsum_scaled);
            } else {
                memcpy(bsum_w_scaled, bsum_w, sizeof(real_t) * ks * dimA);
                for (size_t i = 0; i < dimA * ks; i++) bsum_w_scaled[i] *= -step_size;
            }
            real_t cnst_div = 1. / (1. + 2. * l2_reg * step_size);
            oracle_pg_iteration(A, B, Xr, Xr_indptr, Xr_indices, dimA, ks, cnst_div, bsum_scaled,
                                bsum_w_scaled, step_size, w_mult, maxupd, nthreads);
            step_size *= 0.5;
        }
    } else if (method == M_CG) {
        oracle_cg_iteration(A, B, Xr, Xr_indptr, Xr_indices, dimA, ks, limit_step, Bsum, l2_reg, w_mult,
                            maxupd * niter, bsum_w, nthreads);
    } else {
        oracle_tncg_iteration(A, B, reuse_mean, Xr, Xr_indptr, Xr_indices, dimA, ks, Bsum, l2_reg, w_mult,
                              (int)maxupd, 0, bsum_w, nthreads);
    }
    free(bsum_w); free(bsum_w_scaled); free(bsum_scaled);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Serving-side helpers                                                                       */
/* ------------------------------------------------------------------------------------------ */
/* ref: src/pred.c:42-64 (predict_multiple) */
ORC_API void oracle_predict_multiple(real_t *out, const real_t *A, const real_t *B, const sparse_ix *ixA,
                                     const sparse_ix *ixB, size_t n, int k, int nthreads)
{
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (size_t i = 0; i < n; i++)
        out[i] = vdot(k, A + (size_t)ixA[i] * (size_t)k, B + (size_t)ixB[i] * (size_t)k);
}

/* ref: src/topN.c:112-284 (topN).  The reference's three code paths (include list, large exclude list, gemv +
   partial argsort) all compute the same thing: the n_top best-scoring candidates in descending score order.  The
   order among EQUAL scores is whatever its qsort / quickselect leaves, i.e. unspecified; this restatement breaks
   ties by ascending index.  Return codes as at ref :126-130. */
typedef struct { real_t s; sparse_ix j; } scored_t;
static int cmp_scored(const void *a, const void *b)
{
    const scored_t *x = (const scored_t *)a, *y = (const scored_t *)b;
    if (x->s != y->s) return (x->s < y->s) ? 1 : -1;
    return (x->j > y->j) - (x->j < y->j);
}
ORC_API int oracle_topn(const real_t *a_vec, const real_t *B, int k, const sparse_ix *include_ix, size_t n_include,
                        const sparse_ix *exclude_ix, size_t n_exclude, sparse_ix *outp_ix, real_t *outp_score,
                        size_t n_top, size_t n)
{
    if (n_include == 0) include_ix = NULL;
    if (n_exclude == 0) exclude_ix = NULL;
    if (include_ix != NULL && exclude_ix != NULL) return 2;
    if (n_top == 0) return 2;
    if (n_exclude > n - n_top) return 2;
    if (n_include > n) return 2;
    size_t n_cand = include_ix ? n_include : n;
    if (n_top > n_cand) return 2;
    scored_t *c = (scored_t *)malloc(sizeof(scored_t) * n_cand);
    char *skip = (char *)calloc(n, 1);
    if (exclude_ix) for (size_t i = 0; i < n_exclude; i++) skip[exclude_ix[i]] = 1;
    size_t m = 0;
    for (size_t i = 0; i < n_cand; i++) {
        sparse_ix j = include_ix ? include_ix[i] : (sparse_ix)i;
        if (!include_ix && skip[j]) continue;
        c[m].j = j;
        c[m].s = vdot(k, a_vec, B + (size_t)j * (size_t)k);
        m++;
    }
    qsort(c, m, sizeof(scored_t), cmp_scored);
    for (size_t i = 0; i < n_top; i++) {
        outp_ix[i] = c[i].j;
        if (outp_score) outp_score[i] = c[i].s;
    }
    free(c); free(skip);
    return 0;
}
