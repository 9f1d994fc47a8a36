"""Shared builders for the test-suite (inputs only; the checker lives in oracle/)."""
import numpy as np
import scipy.sparse as sp

from poismf_amd import harness, synth


def checker(use_float, method="cg"):
    """The oracle flavour GPU results are judged against.  PG: the plain flavour (literal left-to-right arithmetic) --
    PG has no line search, agrees to 1e-12 / 1e-5 with any summation order, and its overflow cases (x / 0 = inf feeding
    0 * inf = NaN, which the reference has no guard against) must come out entry for entry as IEEE arithmetic gives them,
    not as a BLAS that skips `y += 0 * x` does.  CG / TNCG, preferred: oracle/_ref/liboracle_blas_*.so -- this repo's
    restatement with its k-length sums routed through the BLAS the compiled reference links, which reproduces the
    compiled reference BIT FOR BIT (tests/test_oracle_vs_ref.py) and travels to the GPU box with the snapshot.  The
    solvers' line searches make discrete decisions on rounding-level differences: on BASELINE-sized rows the plain
    left-to-right flavour ends some rows elsewhere than the reference does (C5, fp64 TNC, 29-nonzero item rows: line
    search failure at f = 15484 where the reference converges at f = 12861 -- and the GPU lands on 12861.306156, the
    reference's value, to 1e-10).  Fallback where _ref is absent: the flavour with fused multiply-add in y += a x."""
    from oracle import bindings
    if method == "pg":
        return bindings.Oracle(use_float)
    try:
        if bindings.ref_available(use_float):
            return bindings.Oracle(use_float, blas_flavour=True)
    except OSError:
        pass
    return bindings.Oracle(use_float, fma_axpy=True)


def dtype_of(use_float):
    return np.float32 if use_float else np.float64


def c1_problem(use_float, k=5, seed=1):
    """BASELINE config C1: the reference's README data."""
    coo = synth.readme_coo()
    csr, csc = harness.process_data(coo, use_float)
    A0, B0 = harness.initialize_matrices(coo.shape[0], coo.shape[1], k, use_float, seed)
    return csr, csc, A0, B0


def small_problem(dimA, dimB, nnz, k, use_float, seed=0, empty_rows=(), empty_cols=(), powerlaw=False):
    """Random count matrix with optional all-empty rows/columns and a power-law column profile."""
    rng = np.random.default_rng(seed)
    row = rng.integers(0, dimA, nnz)
    if powerlaw:
        p = (np.arange(dimB) + 1.0) ** -0.9
        col = rng.choice(dimB, size=nnz, p=p / p.sum())
    else:
        col = rng.integers(0, dimB, nnz)
    val = 1.0 + np.floor(rng.gamma(1.0, 1.0, nnz))
    keep = ~np.isin(row, list(empty_rows)) & ~np.isin(col, list(empty_cols))
    coo = sp.coo_matrix((val[keep], (row[keep], col[keep])), shape=(dimA, dimB))
    csr, csc = harness.process_data(coo, use_float)
    A0, B0 = harness.initialize_matrices(dimA, dimB, k, use_float, seed + 1)
    return csr, csc, A0, B0


def random_row(k, nnz, dimF, use_float, seed=0, l1=0.0, scale=1.0):
    """One row sub-problem: opposing factor F, a feasible start a, Bsum = colsum(F) + l1."""
    rng = np.random.default_rng(seed)
    dt = dtype_of(use_float)
    F = (scale * (0.3 + rng.uniform(0, 0.01, (dimF, k)))).astype(dt)
    a = (0.3 + rng.uniform(0, 0.01, k)).astype(dt)
    xind = np.sort(rng.choice(dimF, size=min(nnz, dimF), replace=False)).astype(np.uint64)
    xval = (1.0 + np.floor(rng.gamma(1.0, 1.0, len(xind)))).astype(dt)
    bsum = (F.astype(np.float64).sum(0) + l1).astype(dt)
    return F, a, bsum, xval, xind


def rel_err(x, ref):
    x = np.asarray(x, np.float64)
    ref = np.asarray(ref, np.float64)
    denom = np.maximum(np.abs(ref), 1e-300)
    return float(np.max(np.abs(x - ref) / denom)) if x.size else 0.0


def scaled_err(x, ref):
    """max |x - ref| / max |ref|: element-wise error relative to the magnitude of the whole array
    (robust for entries that are exactly or nearly zero)."""
    x = np.asarray(x, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.max(np.abs(x - ref)) / max(float(np.max(np.abs(ref))), 1e-300))


def half_objective(M, F, data, indices, indptr, bsum, l2, w=1.0):
    """sum over rows of the row sub-problem objective  bsum.a + l2 |a|^2 - w sum_j x_j log(a.F_j)  in fp64
    (rows with no nonzeros contribute their regularisation terms only)."""
    M64, F64 = np.asarray(M, np.float64), np.asarray(F, np.float64)
    rows = np.repeat(np.arange(M64.shape[0]), np.diff(indptr.astype(np.int64)))
    pred = np.einsum("ij,ij->i", M64[rows], F64[indices.astype(np.int64)])
    ll = float(np.sum(np.asarray(data, np.float64) * np.log(pred)))
    return float(M64 @ np.asarray(bsum, np.float64)).__float__() if False else \
        float((M64 @ np.asarray(bsum, np.float64)).sum() + l2 * (M64 ** 2).sum() - w * ll)


def frac_rows_close(X, ref, tol):
    """fraction of rows whose max abs error is within tol * max|ref| (for chaotic fp32 solvers)."""
    X, ref = np.asarray(X, np.float64), np.asarray(ref, np.float64)
    scale = max(float(np.max(np.abs(ref))), 1e-300)
    return float(np.mean(np.max(np.abs(X - ref), axis=1) <= tol * scale))


def check_topn(a_vec, B, idx, scores, include_ix, exclude_ix, n_top, rtol):
    """A top-N answer is right if (i) it lists n_top distinct admissible items, (ii) in non-increasing score order,
    (iii) the reported scores are the items' scores, and (iv) no admissible item left out beats the last one listed
    by more than rounding -- which is all the reference promises (equal scores come back in qsort's order)."""
    true = np.asarray(B, np.float64) @ np.asarray(a_vec, np.float64)
    idx = np.asarray(idx, np.int64)
    assert len(idx) == n_top and len(set(idx.tolist())) == n_top
    cand = np.asarray(include_ix, np.int64) if len(include_ix) else np.setdiff1d(np.arange(B.shape[0]), np.asarray(exclude_ix, np.int64))
    assert np.isin(idx, cand).all()
    tol = rtol * max(float(np.max(np.abs(true))), 1e-300)
    s = true[idx]
    assert np.all(s[:-1] >= s[1:] - tol)
    if scores is not None and len(scores):
        assert np.max(np.abs(np.asarray(scores, np.float64) - s)) <= tol
    left_out = np.setdiff1d(cand, idx)
    if len(left_out):
        assert true[left_out].max() <= s[-1] + tol


def _reorder_rows(cs, mode, rng):
    """the same CSR / CSC matrix with the nonzeros of every row stored in another order (reversed / shuffled)"""
    data, idx, ptr = cs[0].copy(), cs[1].copy(), cs[2]
    p = ptr.astype(np.int64)
    for r in range(len(p) - 1):
        a, b = p[r], p[r + 1]
        if b - a > 1:
            o = np.arange(b - a)[::-1] if mode == "rev" else rng.permutation(b - a)
            data[a:b] = data[a:b][o]
            idx[a:b] = idx[a:b][o]
    return data, idx, ptr


def tncg_yardstick(use_float, csr, csc, A0, B0, args, nthreads=8):
    """(A, B of the checker, objective of the checker, self-variance): how far the REFERENCE's arithmetic moves on this very problem when only
    the order of its sums changes.  TNC stops a row when its objective moves by less than ftol = 1e-4 of itself (ref: src/poismf.c:383-391,
    src/tnc.c:909-915) and takes its Hessian products from finite differences of step ~1.5e-8, so rounding-level differences decide where a
    row ends; how much of that reaches the TOTAL is a property of the problem (how much of it a few long rows carry).  Six runs on the CPU:
    the two summation flavours of the k-length sums -- the BLAS-routed restatement that reproduces the compiled reference bit for bit
    (`checker`, the first run: what GPU results are judged against) and the plain left-to-right loops -- each on the matrix as given, with
    every row's nonzeros stored in reverse order, and shuffled (the same matrix: the ABI does not ask for sorted rows; this is
    the kind of difference a device that adds a row's nonzeros in tree order has).  self-variance = the largest relative distance of the
    five others' objectives from the checker's.  SURVEY 8c's end-to-end bound for TNCG fp64 is 1e-5: `tncg_bound` lets a test exceed it
    only by a multiple of this measured spread, and the tests print both numbers."""
    from oracle import bindings
    rng = np.random.default_rng(0)
    first, objs = None, []
    for lib in (checker(use_float, "tncg"), bindings.Oracle(use_float)):
        for mode in (None, "rev", "perm"):
            c1 = csr if mode is None else _reorder_rows(csr, mode, rng)
            c2 = csc if mode is None else _reorder_rows(csc, mode, rng)
            A, B = A0.copy(), B0.copy()
            rc = lib.run_poismf(A, c1[0], c1[2], c1[1], B, c2[0], c2[2], c2[1], args["l2_reg"], args["l1_reg"], args["w_mult"],
                                args["step_size"], "tncg", args["limit_step"], args["niter"], args["maxupd"], args["early_stop"],
                                args["reuse_prev"], True, nthreads)
            assert rc == 0
            objs.append(harness.poisson_objective(A, B, csr, args["l2_reg"], args["l1_reg"], args["w_mult"]))
            if first is None:
                first = (A, B)
    return first[0], first[1], objs[0], max(abs(o - objs[0]) for o in objs[1:]) / abs(objs[0])


def tncg_bound(self_variance, floor=1e-5):
    """SURVEY 8c's 1e-5, or twice the largest distance the reference's own arithmetic shows on the same rows under another order of its
    sums (tncg_yardstick: five alternatives), whichever is larger"""
    return max(floor, 2.0 * self_variance)
