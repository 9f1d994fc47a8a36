/*
 * poismf_hip.h -- C-ABI of the MI355X (gfx950) implementation of poismf's alternating factor-update
 * hot path.  Two shared libraries export exactly these symbols, as the reference builds its core
 * twice (ref: setup.py:225-243, src/poismf.h:91-109):
 *
 *     libpoismf_hip_d.so   real_t = double
 *     libpoismf_hip_f.so   real_t = float      (compile this header with -DUSE_FLOAT)
 *     libpoismf_hip_r.so   real_t = double, sparse_ix = int: the reference's R ABI (compile with -D_FOR_R,
 *                          ref: src/poismf.h:75-89; same row kernels as libpoismf_hip_d.so)
 *
 * sparse_ix is size_t otherwise (the reference's C/Python ABI, ref: src/poismf.h:76).  No torch / HIP types
 * appear in any signature: pointers and sizes only.  All "ref:" citations are relative to the
 * reference tree (david-cortes/poismf).
 *
 * There is NO CPU fallback: every entry point fails (non-zero return) if no HIP device is usable.
 */
#ifndef POISMF_HIP_H
#define POISMF_HIP_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ref: src/poismf.h:75-109.  -D_FOR_R selects the R ABI: int indices, double only (libpoismf_hip_r.so). */
#ifdef _FOR_R
  #undef USE_FLOAT
#endif
#ifndef real_t
  #ifdef USE_FLOAT
    #define real_t float
  #else
    #define real_t double
  #endif
#endif
#ifndef sparse_ix
  #ifdef _FOR_R
    #define sparse_ix int
  #else
    #define sparse_ix size_t
  #endif
#endif
#if defined(__GNUC__) || defined(__clang__)
  #define POISMF_HIP_API __attribute__((visibility("default")))
#else
  #define POISMF_HIP_API
#endif

/* ref: src/poismf.h:225  typedef enum Method {tncg = 1, cg = 2, pg = 3} Method; */
typedef enum poismf_hip_method { POISMF_TNCG = 1, POISMF_CG = 2, POISMF_PG = 3 } poismf_hip_method;

/* ---------------------------------------------------------------------------------------------
 * 1. Drop-in entry point.
 *
 * Replaces: run_poismf, ref: src/poismf.h:226-233 (prototype), src/poismf.c:435-632 (body).
 * Same name, same argument order (NOTE indptr before indices), same in-place A/B semantics, same
 * return codes: 0 ok, 1 out of memory (host or device; also printed to stderr as the reference
 * does, ref: src/poismf.c:501), 2 interrupted by SIGINT (ref: src/poismf.c:618-630).
 * `method` is the reference's enum passed as int (1 tncg, 2 cg, 3 pg).
 * `nthreads` is accepted and ignored (the row loop runs on the GPU).
 * A, B, X* are HOST pointers owned by the caller; A and B are overwritten with the result.
 * The device used is HIP device $POISMF_HIP_DEVICE (default 0).
 * ------------------------------------------------------------------------------------------- */
POISMF_HIP_API int run_poismf(
    real_t *A, real_t *Xr, sparse_ix *Xr_indptr, sparse_ix *Xr_indices,
    real_t *B, real_t *Xc, sparse_ix *Xc_indptr, sparse_ix *Xc_indices,
    const size_t dimA, const size_t dimB, const size_t k,
    const real_t l2_reg, const real_t l1_reg, const real_t w_mult, real_t step_size,
    const int method, const bool limit_step, const size_t numiter, const size_t maxupd,
    const bool early_stop, const bool reuse_prev,
    const bool handle_interrupt, const int nthreads);

/* ---------------------------------------------------------------------------------------------
 * 1b. Factors for new rows with B fixed.
 *
 * Replaces: factors_multiple, ref: src/poismf.h:270-280 (prototype), src/pred.c:66-199 (body); called by the
 * Python transform() through poismf_c_wrapper.pxi:147-199.  Same name, argument order and meaning: A [dimA x k]
 * is output only (host), B / Bsum (with l1 already added) / Amean come from the fitted model, Xr* is the CSR of
 * the new rows.  It is one half-sweep of the same row kernels with the column sums supplied by the caller.
 * Returns 0, or 1 when out of memory / no device.
 * ------------------------------------------------------------------------------------------- */
POISMF_HIP_API int factors_multiple(
    real_t *A, real_t *B, real_t *Bsum, real_t *Amean,
    real_t *Xr, sparse_ix *Xr_indptr, sparse_ix *Xr_indices,
    int k, size_t dimA,
    real_t l2_reg, real_t w_mult,
    real_t step_size, size_t niter, size_t maxupd,
    int method, bool limit_step, bool reuse_mean,
    int nthreads);

/* ---------------------------------------------------------------------------------------------
 * 1c. COO -> CSR + CSC on the device.
 *
 * Replaces: the SciPy conversions of PoisMF._process_data, ref: poismf/__init__.py:404-414 (coo.tocsr() and
 * coo.tocsc(): duplicate (i,j) entries summed, indices sorted within each row / column, then cast to real_t /
 * size_t).  There is no C function for this in the reference; the entry point takes what _process_data holds
 * (host triplets) and fills what run_poismf takes.  Output arrays are caller-allocated with capacity n
 * (values, indices) and dim + 1 (indptr); *nnz_out receives the number of distinct (i,j).  Requires n < 2^32.
 * Returns 0, or 1 when out of memory / no device.
 * ------------------------------------------------------------------------------------------- */
POISMF_HIP_API int poismf_hip_coo_to_csr_csc(
    const sparse_ix *row, const sparse_ix *col, const real_t *val, size_t n, size_t dimA, size_t dimB,
    real_t *csr_val, sparse_ix *csr_indices, sparse_ix *csr_indptr,
    real_t *csc_val, sparse_ix *csc_indices, sparse_ix *csc_indptr, size_t *nnz_out);

/* ---------------------------------------------------------------------------------------------
 * 1d. Serving-side helpers.
 *
 * Replaces: predict_multiple, ref: src/poismf.h:250-257 (prototype), src/pred.c:42-64;
 *           topN,             ref: src/poismf.h:240-247 (prototype), src/topN.c:112-284.
 * Same names, argument order and return codes (topN: 0 ok, 1 out of memory / no device, 2 invalid combination of
 * include / exclude / n_top as at ref src/topN.c:126-130).  Among equal scores topN returns ascending indices (the
 * reference leaves that order to qsort).  Host pointers in and out; the factors are copied to the device per call.
 * ------------------------------------------------------------------------------------------- */
POISMF_HIP_API void predict_multiple(
    real_t *out, real_t *A, real_t *B, sparse_ix *ixA, sparse_ix *ixB, size_t n, int k, int nthreads);
POISMF_HIP_API int topN(
    real_t *a_vec, real_t *B, int k,
    sparse_ix *include_ix, size_t n_include,
    sparse_ix *exclude_ix, size_t n_exclude,
    sparse_ix *outp_ix, real_t *outp_score,
    size_t n_top, size_t n, int nthreads);

/* ---------------------------------------------------------------------------------------------
 * 2. Device-resident session: the same path with X, A and B kept in HBM between calls, one
 *    half-sweep per call.  This is what bench.py times (inputs already resident) and what the
 *    one-process-per-GPU driver uses: each rank owns a contiguous range of A rows and of B rows,
 *    both factors are replicated, and the caller all-gathers the updated shard between halves.
 *
 * Replaces, per call: sum_by_cols + l1 + PG pre-scaling (ref: src/poismf.c:77-83, :512-526,
 * :562-577) and pg_iteration / cg_iteration / tncg_iteration (ref: src/poismf.c:139-188, :275-322,
 * :324-404) for the rows of the shard.
 * ------------------------------------------------------------------------------------------- */
typedef struct poismf_hip_session poismf_hip_session;

/* Creates a session on HIP device `device`.  Xr* / Xc* are HOST CSR / CSC arrays of the WHOLE matrix
 * (size_t indices, as in run_poismf); only rows [rowA_begin,rowA_end) of the CSR and rows
 * [rowB_begin,rowB_end) of the CSC (= columns of X) are uploaded.  Pass 0,dimA / 0,dimB for a
 * single-GPU session.  `stream` is a hipStream_t passed as void*; all work of this session is enqueued on it.
 * NULL does NOT mean the legacy default stream: the session then creates and owns a non-blocking stream of its
 * own, which is not ordered against work the caller enqueues elsewhere -- a caller that touches the factors
 * through the device pointers below must either pass its own stream here or order its work against
 * poismf_hip_session_stream().  Returns 0, or 1 when out of memory / no device. */
POISMF_HIP_API int poismf_hip_session_create(
    poismf_hip_session **out, int device, void *stream,
    const real_t *Xr, const sparse_ix *Xr_indptr, const sparse_ix *Xr_indices,
    const real_t *Xc, const sparse_ix *Xc_indptr, const sparse_ix *Xc_indices,
    size_t dimA, size_t dimB, size_t k,
    size_t rowA_begin, size_t rowA_end, size_t rowB_begin, size_t rowB_end);

/* The same session built from HOST COO triplets (what PoisMF._process_data holds, ref: poismf/__init__.py:404-414):
 * both orientations are converted on the device (duplicates summed, indices sorted, section 1c) and stay there -- the
 * CSR / CSC never exist in host memory.  Only triplets whose row lies in [rowA_begin,rowA_end) enter the CSR shard and
 * only those whose column lies in [rowB_begin,rowB_end) the CSC shard.  Requires n < 2^32.  Returns 0; 1 out of memory / no
 * device; 3 when a row or column index lies outside the matrix (it would become a gather offset into the factors). */
POISMF_HIP_API int poismf_hip_session_create_coo(
    poismf_hip_session **out, int device, void *stream,
    const sparse_ix *row, const sparse_ix *col, const real_t *val, size_t n,
    size_t dimA, size_t dimB, size_t k,
    size_t rowA_begin, size_t rowA_end, size_t rowB_begin, size_t rowB_end);

POISMF_HIP_API void poismf_hip_session_destroy(poismf_hip_session *s);

/* The stream this session enqueues its work on (the one given at creation, or the one it created), as void*. */
POISMF_HIP_API void *poismf_hip_session_stream(poismf_hip_session *s);

/* Device pointers to the session-owned, replicated factors: A is [dimA x k], B is [dimB x k],
 * row-major real_t (allocations carry 16 bytes of slack because rows are gathered in 16-byte
 * slots).  The caller may wrap them (e.g. as torch tensors) to run collectives on them.
 * For small k the session also keeps a line-padded copy of each factor for its gathers; a
 * session whose shard is the whole factor refreshes that copy from its own row kernels and
 * must be told about outside writes: call poismf_hip_session_factors_dirty(s, which) (which = 0: B was
 * written, 1: A) after every write through a pointer obtained earlier (calling the getter again or
 * set_factors has the same effect).  Sessions with partial shards re-derive the copy from the compact
 * factor before every half-sweep in any case. */
POISMF_HIP_API real_t *poismf_hip_session_A(poismf_hip_session *s);
POISMF_HIP_API real_t *poismf_hip_session_B(poismf_hip_session *s);
POISMF_HIP_API void poismf_hip_session_factors_dirty(poismf_hip_session *s, int which);

/* Host <-> device copies of the full factors (synchronous with respect to the session stream).
 * get_factors (like run_poismf and poismf_hip_session_run) also returns 1 when a half-sweep since the last check lost a
 * row launch that shares rows between CUs (fp64 CG, 385-2048 nonzeros; the kernels give up after ~1 s without an answer
 * from a team member instead of hanging the device): the factors are then not to be used. */
POISMF_HIP_API int poismf_hip_session_set_factors(poismf_hip_session *s, const real_t *A_host, const real_t *B_host);
POISMF_HIP_API int poismf_hip_session_get_factors(poismf_hip_session *s, real_t *A_host, real_t *B_host);

/* Hyper-parameters of the alternation; same meaning as the run_poismf arguments. */
typedef struct poismf_hip_params {
    real_t l2_reg, l1_reg, w_mult, step_size;
    int method;          /* 1 tncg, 2 cg, 3 pg */
    int limit_step;
    size_t maxupd;
    int early_stop, reuse_prev;
} poismf_hip_params;

/* One half-sweep over this session's shard.  which = 0: update B rows [rowB_begin,rowB_end) against
 * the full A (the reference's first half, ref: src/poismf.c:512-556); which = 1: update A rows against
 * the full B (ref: src/poismf.c:562-603).  The column sums of the opposing factor are recomputed on
 * the device from the replicated copy, so no k-vector collective is needed.  `step_size` is the step
 * of THIS half and `cnst_div` the PG divisor (the caller keeps the reference's step schedule: cnst_div from
 * the step before halving, A half run with the halved step -- quirk Q6; the double scaling of the column
 * sums on the A half -- quirk Q1 -- is applied inside).  If n_unchanged != NULL (TNCG early stop) it receives
 * the number of shard rows whose update moved by <= 1e-4 in squared norm (ref: src/poismf.c:393-396);
 * reading it synchronises the stream.  Asynchronous otherwise.  Returns 0 or 1. */
POISMF_HIP_API int poismf_hip_half_sweep(poismf_hip_session *s, int which, const poismf_hip_params *p,
                          real_t step_size, real_t cnst_div, size_t *n_unchanged);

/* Segments (multi-GPU overlap of the shard exchange with compute).  set_segments cuts the
 * session's shard of half `which` into `nseg` contiguous row ranges of equal row counts (rows begin + n j / nseg ..
 * begin + n (j + 1) / nseg), each sorted and binned on its own, and returns the number of segments (< 0 on error);
 * segment_rows reports a segment's global row range; half_sweep_segment runs ONE segment: segment 0 also computes the
 * column sums and resets the early-stop counter, the call that passes n_unchanged (the last segment) reads it.  Running
 * segments 0 .. nseg-1 in order is bit-identical to one poismf_hip_half_sweep (a row's arithmetic depends on its
 * length class only).  The caller orders its exchange of segment j's rows after that call on the session stream. */
POISMF_HIP_API int poismf_hip_session_set_segments(poismf_hip_session *s, int which, int nseg);
POISMF_HIP_API int poismf_hip_session_segment_rows(poismf_hip_session *s, int which, int seg, size_t *row_begin, size_t *row_end);
POISMF_HIP_API int poismf_hip_half_sweep_segment(poismf_hip_session *s, int which, const poismf_hip_params *p,
                          real_t step_size, real_t cnst_div, int seg, size_t *n_unchanged);

/* run_poismf's outer loop (ref: src/poismf.c:506-608: B half, PG step halving, A half, TNCG early stop, SIGINT
 * handling and return codes 0 / 1 / 2 exactly as in section 1) on a session that already holds X and the starting
 * factors; `p->step_size` is the initial step.  PoisMF.fit uses it with poismf_hip_session_create_coo so that the
 * converted CSR / CSC never leave the device. */
POISMF_HIP_API int poismf_hip_session_run(poismf_hip_session *s, const poismf_hip_params *p, size_t numiter, int handle_interrupt);

/* Wall-clock (HIP events on the session stream) of the row-update kernels launched by half-sweeps
 * since profiling was switched on: total milliseconds and number of kernel launches, per half.
 * Synchronises the stream. */
POISMF_HIP_API void poismf_hip_session_profile(poismf_hip_session *s, int enable);
POISMF_HIP_API int poismf_hip_session_kernel_time(poismf_hip_session *s, int which, double *total_ms, size_t *launches);

/* While profiling is enabled the row kernels also count, per half (which = 0: B, 1: A) and since the last
 * poismf_hip_session_profile() call, the passes they made over rows' gathered tiles (one per gradient / function
 * evaluation of the inner solver, ref src/poismf.c:126-133, :194-273) and the sum over rows of passes x nonzeros --
 * SURVEY.md 8(d)'s pass-weighted effective traffic is nnz_passes * k * sizeof(real_t). */
POISMF_HIP_API int poismf_hip_session_eval_stats(poismf_hip_session *s, int which, unsigned long long *tile_passes, unsigned long long *nnz_passes);

/* Diagnostic: the kernels take log() of the predictions in double (as the reference's C does even in its float build,
 * ref src/poismf.c:199, :268) with their own implementation of the fdlibm algorithm instead of the device library's.
 * This runs both on n sample arguments on the current device and reports the largest distance in ulps and the number
 * of special arguments (+-0, -1, +-inf, NaN) on which they disagree.  Returns 0 on success. */
POISMF_HIP_API int poismf_hip_selftest_log(size_t n, unsigned long long *worst_ulp, unsigned *mismatched_specials);

/* Serving from the session's resident factors: the same results as predict_multiple
 * (ref: src/pred.c:42-64) and as topN with a_vec = row `user` of A (ref: src/topN.c:112-284), without the per-call copy
 * of the factors that the host-pointer drop-ins of section 1d make.  Index arrays are host arrays.  Return 0, 1 (device
 * error / out of memory) or 2 (an index out of range, or topN's invalid combinations, ref src/topN.c:126-130). */
POISMF_HIP_API int poismf_hip_session_predict(poismf_hip_session *s, const sparse_ix *ixA, const sparse_ix *ixB, size_t n, real_t *out);
POISMF_HIP_API int poismf_hip_session_topn(poismf_hip_session *s, size_t user,
                          const sparse_ix *include_ix, size_t n_include, const sparse_ix *exclude_ix, size_t n_exclude,
                          sparse_ix *outp_ix, real_t *outp_score, size_t n_top);

/* Which row-kernel instances the most recent half-sweep of half `which` launched, as text ("kernel<instance> rows=N;"
 * per launch), NUL-terminated and truncated to cap bytes; returns the untruncated length.  Reporting only. */
POISMF_HIP_API size_t poismf_hip_session_plan(poismf_hip_session *s, int which, char *buf, size_t cap);

/* Profiling sessions (poismf_hip_session_profile(s, 1)) also bracket every row-bin launch with events on the stream it
 * is issued on.  This returns, for half `which`, "kernel<instance> rows=R nnz=Z calls=C ms=T;" per distinct launch since
 * profiling was switched on (T = summed milliseconds over the C calls; Z = nonzeros of the rows the launch covers, i.e.
 * what its algorithmic bytes are computed from).  Same buffer convention as poismf_hip_session_plan.  Reporting only. */
POISMF_HIP_API size_t poismf_hip_session_launch_profile(poismf_hip_session *s, int which, char *buf, size_t cap);

/* Profiling sessions also record what every row's inner solver decided in the most recent half-sweep of half `which`:
 * out[2 r] = iterations | rc << 24, out[2 r + 1] = evaluations, for local row r -- the numbers the reference's
 * minimize_nonneg_cg (niter, nfeval; ref src/nonnegcg.c:177-189) and tnc (nfeval, niter, rc; ref src/tnc.c:251-260) return and
 * cg_iteration / tncg_iteration discard.  Testing aid: pins the solvers' decisions, not only their results. */
POISMF_HIP_API int poismf_hip_session_decisions(poismf_hip_session *s, int which, unsigned *out, size_t nrows);
/* The same, summed over the rows of the shard: out[0] = sum of iterations, out[1] = sum of evaluations, out[2] = sum of
 * nnz x iterations, out[3] = sum of nnz x evaluations -- what a flop count of the reference's arithmetic for the same decisions
 * needs (a gradient is 4k+1 flops per nonzero, a function value 2k+L; ref src/poismf.c:126-133, :194-208).
 * Returns 1 when the session is not profiling. */
POISMF_HIP_API int poismf_hip_session_decision_stats(poismf_hip_session *s, int which, unsigned long long *out);
/* factors_multiple (below / ref src/pred.c:66-199) that also returns those two words per row. */
POISMF_HIP_API int poismf_hip_factors_multiple_decisions(real_t *A, real_t *B, real_t *Bsum, real_t *Amean, real_t *Xr,
                          sparse_ix *Xr_indptr, sparse_ix *Xr_indices, int k, size_t dimA, real_t l2_reg, real_t w_mult,
                          real_t step_size, size_t niter, size_t maxupd, int method, bool limit_step, bool reuse_mean,
                          unsigned *decisions);

/* The first stage of a half-sweep's column sums, shared between the ranks of a multi-GPU run (the sum itself: ref
 * src/poismf.c:77-83).  The sum over the FIXED factor of half `which` (A for which = 0, B for which = 1) is cut into
 * poismf_hip_session_colsum_blocks() blocks whose partial sums depend on the block number alone: a rank computes blocks [b_lo, b_hi) into
 * the session's partial array (poismf_hip_session_partials: [blocks x k] real_t in device memory), receives the others from its peers
 * into the same array, and declares it complete (poismf_hip_session_partials_ready): the next half-sweep (or its segment 0) then runs
 * only the fixed-order second stage.  Bit for bit the unsharded sum. */
POISMF_HIP_API int poismf_hip_session_colsum_blocks(poismf_hip_session *s, int which);
POISMF_HIP_API int poismf_hip_session_colsum_partial(poismf_hip_session *s, int which, int b_lo, int b_hi);
POISMF_HIP_API real_t *poismf_hip_session_partials(poismf_hip_session *s);
POISMF_HIP_API void poismf_hip_session_partials_ready(poismf_hip_session *s);

/* Nothing survives run_poismf by default: every device array the call allocated is freed before it returns, as the reference frees
 * its scratch (ref src/poismf.c:610-619).  A caller that fits repeatedly on matrices of one shape can OPT IN to keeping released
 * device arrays of 1 MB and more for the next request of the same size on the same device (repeated fits then allocate nothing and
 * their uploads run over two DMA queues at full rate): environment POISMF_HIP_DEVICE_CACHE_MB=<MB> or
 * poismf_hip_set_device_cache_mb(MB) at run time (returns the previous limit; lowering it frees what no longer fits; the limit is
 * per loaded flavour of the library).  poismf_hip_release_cache() hands whatever is kept back to the driver. */
POISMF_HIP_API void poismf_hip_release_cache(void);
POISMF_HIP_API size_t poismf_hip_set_device_cache_mb(size_t mb);

/* Testing aid (G1): the device's own objective / gradient wrappers at `point`, for every row of a CSR, through whichever row engine a CG
 * half-sweep would use for a row of that length.  which = 0: calc_fun_single + calc_grad_single[_w] (ref src/poismf.c:194-240);
 * which = 1: calc_fun_and_grad (ref src/poismf.c:242-273; no l2 term in f).  G [dimA x k] receives the gradients, f [dimA] the values. */
POISMF_HIP_API int poismf_hip_debug_row_eval(real_t *G, double *f, real_t *B, real_t *Bsum, real_t *point, real_t *Xr,
                          sparse_ix *Xr_indptr, sparse_ix *Xr_indices, int k, size_t dimA, real_t l2_reg, real_t w_mult, int which);

/* Number of nonzeros held by this session for half `which` (shard only). */
POISMF_HIP_API size_t poismf_hip_session_nnz(poismf_hip_session *s, int which);

#ifdef __cplusplus
}
#endif
#endif /* POISMF_HIP_H */
