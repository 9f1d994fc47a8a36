"""Pins the oracle: the C restatement (oracle/poismf_oracle.c) against the REAL reference compiled in
place from /root/reference/src (oracle/_ref, built by `make -C oracle ref`).

Levels follow SURVEY.md section 8c: G1 primitives, G2 single-row solvers, G3 half-sweeps, G4 full
run_poismf, G5 edge cases.  Tolerances come from the reference's own self-variance across BLAS
builds (SURVEY.md section 7 / BASELINE.md section 2): the restatement uses plain left-to-right
k-length sums whereas the compiled reference calls SciPy's OpenBLAS (SIMD summation order + FMA).
"""
import numpy as np
import pytest

from oracle import bindings
from poismf_amd import harness
from tests import helpers as H

pytestmark = pytest.mark.skipif(not (bindings.ref_available(False) and bindings.ref_available(True)),
                                reason="compiled reference (oracle/_ref) not present")


# Two flavours of the SAME restatement source are pinned:
#   "plain": oracle/liboracle_*.so, portable left-to-right sums, -ffp-contract=off.  This is THE oracle the
#            GPU tests use; it agrees with the reference to the reference's own BLAS-to-BLAS variance.
#   "blas":  oracle/_ref/liboracle_blas_*.so, k-length sums routed through the reference's own BLAS and
#            built with the reference's flags.  It must agree with the reference BIT FOR BIT, which pins
#            every branch of the three solvers and the outer loop (no tolerance to hide behind).
_EXACT = {"on": False}


@pytest.fixture(scope="module", params=[(False, False), (True, False), (False, True), (True, True)],
                ids=["f64-plain", "f32-plain", "f64-blas", "f32-blas"])
def libs(request):
    is_float, blas = request.param
    _EXACT["on"] = blas
    return bindings.Oracle(is_float, blas_flavour=blas), bindings.Reference(is_float), is_float


def tol(is_float, t64, t32):
    if _EXACT["on"]:
        return 0.0 if False else 1e-300  # bit-exact flavour: nothing but a true zero difference passes
    return t32 if is_float else t64


# ---------------------------------------------------------------- G1
@pytest.mark.parametrize("k,nnz", [(5, 1), (5, 7), (50, 100), (100, 3000), (50, 7)])
@pytest.mark.parametrize("w,l1", [(1.0, 0.0), (10.0, 0.5)])
def test_g1_primitives(libs, k, nnz, w, l1):
    orc, ref, is_float = libs
    F, a, bsum, xval, xind = H.random_row(k, nnz, 4000, is_float, seed=k * 1000 + nnz, l1=l1)
    l2 = 1e3
    t = tol(is_float, 1e-12, 2e-5)
    assert H.scaled_err(orc.calc_grad_pgd(a, F, xval, xind), ref.calc_grad_pgd(a, F, xval, xind)) <= t
    fo, fr = orc.calc_fun_single(a, F, bsum, xval, xind, l2, w), ref.calc_fun_single(a, F, bsum, xval, xind, l2, w)
    assert abs(fo - fr) <= t * abs(fr)
    for weighted in (False, True):
        go = orc.calc_grad_single(a, F, bsum, xval, xind, l2, w, weighted)
        gr = ref.calc_grad_single(a, F, bsum, xval, xind, l2, w, weighted)
        assert H.scaled_err(go, gr) <= t
    (fo, go), (fr, gr) = orc.calc_fun_and_grad(a, F, bsum, xval, xind, l2, w), ref.calc_fun_and_grad(a, F, bsum, xval, xind, l2, w)
    assert abs(fo - fr) <= t * abs(fr)
    assert H.scaled_err(go, gr) <= t


# ---------------------------------------------------------------- G2
@pytest.mark.parametrize("k,nnz", [(5, 7), (50, 100), (100, 300)])
@pytest.mark.parametrize("limit_step", [True, False])
@pytest.mark.parametrize("maxupd", [1, 5])
def test_g2_cg_row(libs, k, nnz, limit_step, maxupd):
    orc, ref, is_float = libs
    F, a, bsum, xval, xind = H.random_row(k, nnz, 2000, is_float, seed=7 * k + nnz)
    xo, fo, nio, nfo, rco = orc.cg_row(a, F, bsum, xval, xind, 1e4, 1.0, maxupd, limit_step)
    xr, fr, nir, nfr, rcr = ref.cg_row(a, F, bsum, xval, xind, 1e4, 1.0, maxupd, limit_step)
    assert (rco, nio) == (rcr, nir)
    if maxupd == 1:
        assert nfo == nfr
        assert H.scaled_err(xo, xr) <= tol(is_float, 1e-10, 1e-3)
    else:
        assert H.scaled_err(xo, xr) <= tol(is_float, 1e-6, 5e-2)
    assert abs(fo - fr) <= tol(is_float, 1e-10, 1e-5) * abs(fr)


@pytest.mark.parametrize("k,nnz", [(5, 7), (50, 100), (100, 300)])
@pytest.mark.parametrize("maxupd", [10, 75, 750])
@pytest.mark.parametrize("reuse", [True, False])
def test_g2_tnc_row(libs, k, nnz, maxupd, reuse):
    orc, ref, is_float = libs
    F, a, bsum, xval, xind = H.random_row(k, nnz, 2000, is_float, seed=11 * k + nnz)
    if not reuse:
        a = np.full_like(a, 1e-3)
    xo, fo, nfo, nio, rco = orc.tnc_row(a, F, bsum, xval, xind, 1e3, 1.0, maxupd)
    xr, fr, nfr, nir, rcr = ref.tnc_row(a, F, bsum, xval, xind, 1e3, 1.0, maxupd)
    # the objective the solver reports is what parity is judged on; the trajectory is
    # chaotic in fp32 (the reference disagrees with itself across BLAS builds by O(1) element-wise)
    if is_float and maxupd == 10 and not _EXACT["on"]:
        assert abs(fo - fr) <= 0.1 * abs(fr)  # truncated mid-descent in fp32: informational only
    else:
        assert abs(fo - fr) <= tol(is_float, 1e-6, 1e-2) * max(abs(fr), 1.0)
    if _EXACT["on"]:
        assert np.array_equal(xo, xr) and (fo, nfo, nio, rco) == (fr, nfr, nir, rcr)
    elif not is_float:
        assert H.scaled_err(xo, xr) < 5e-3
        assert abs(nfo - nfr) <= max(3, 0.1 * nfr)


# ---------------------------------------------------------------- G3
def _half_inputs(is_float, method):
    csr, csc, A0, B0 = H.c1_problem(is_float)
    l2, maxupd, _ = harness.auto_defaults(method, 5)
    return csr, csc, A0, B0, l2, maxupd


def test_g3_pg_half(libs):
    orc, ref, is_float = libs
    csr, csc, A0, B0, l2, maxupd = _half_inputs(is_float, "pg")
    step = 1e-7
    cd = 1.0 / (1.0 + 2.0 * l2 * step)
    out = []
    for lib in (orc, ref):
        A = A0.copy()
        cs = lib.sum_by_cols(B0)
        cs *= -step
        lib.pg_iteration(A, B0, csr[0], csr[2], csr[1], cd, cs, None, step, 1.0, maxupd)
        out.append(A)
    assert H.scaled_err(out[0], out[1]) <= tol(is_float, 1e-12, 1e-5)


@pytest.mark.parametrize("limit_step", [True, False])
def test_g3_cg_half(libs, limit_step):
    orc, ref, is_float = libs
    csr, csc, A0, B0, l2, maxupd = _half_inputs(is_float, "cg")
    out = []
    for lib in (orc, ref):
        B = B0.copy()
        bs = lib.sum_by_cols(A0)
        lib.cg_iteration(B, A0, csc[0], csc[2], csc[1], limit_step, bs, l2, 1.0, maxupd)
        out.append(B)
    bs = orc.sum_by_cols(A0)
    fo = H.half_objective(out[0], A0, csc[0], csc[1], csc[2], bs, l2)
    fr = H.half_objective(out[1], A0, csc[0], csc[1], csc[2], bs, l2)
    if _EXACT["on"] or not is_float:
        assert abs(fo - fr) <= tol(is_float, 1e-10, 0) * abs(fr)
        assert H.scaled_err(out[0], out[1]) <= tol(is_float, 1e-6, 0)
    else:
        # fp32, one half-sweep from the start: the row objective (~l2 |a|^2 ~ 5e3) resolves to ~5e-4 in
        # fp32, so Armijo decisions sit at noise level and the PATH depends on the BLAS summation order
        # (the two agree again at convergence, see test_g4 with the default numiter).  Mid-path only the
        # descent itself is checked here; the bit-exact flavour pins the rest.
        f0 = H.half_objective(B0, A0, csc[0], csc[1], csc[2], bs, l2)
        assert fo < f0 and fr < f0 and abs(fo - fr) <= 2e-2 * abs(fr)


@pytest.mark.parametrize("reuse", [True, False])
def test_g3_tncg_half(libs, reuse):
    orc, ref, is_float = libs
    csr, csc, A0, B0, l2, maxupd = _half_inputs(is_float, "tncg")
    out = []
    for lib in (orc, ref):
        A = A0.copy()
        bs = lib.sum_by_cols(B0)
        lib.tncg_iteration(A, B0, reuse, csr[0], csr[2], csr[1], bs, l2, 1.0, maxupd, True)
        out.append(A)
    if _EXACT["on"]:
        assert np.array_equal(out[0], out[1])
    else:
        # element-wise agreement is informational for TNCG (the reference differs from itself by 5e-3
        # across BLAS builds); the objective reached is what is pinned
        bs = orc.sum_by_cols(B0)
        fo = H.half_objective(out[0], B0, csr[0], csr[1], csr[2], bs, l2)
        fr = H.half_objective(out[1], B0, csr[0], csr[1], csr[2], bs, l2)
        assert abs(fo - fr) <= tol(is_float, 1e-5, 1e-2) * abs(fr)
        if not is_float:
            assert H.scaled_err(out[0], out[1]) < 2e-2


# ---------------------------------------------------------------- G4 / G5
def _run(lib, csr, csc, A0, B0, method, numiter, k=5, **kw):
    l2, maxupd, niter = harness.auto_defaults(method, k)
    A, B = A0.copy(), B0.copy()
    args = dict(l2_reg=l2, l1_reg=0.0, w_mult=1.0, step_size=1e-7, method=method, limit_step=True,
                numiter=niter if numiter == "default" else numiter, maxupd=maxupd, early_stop=True,
                reuse_prev=False)
    args.update(kw)
    rc = lib.run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], **args)
    assert rc == 0
    return A, B, args


def _check_full(orc, ref, is_float, csr, csc, A0, B0, method, numiter, **kw):
    Ao, Bo, args = _run(orc, csr, csc, A0, B0, method, numiter, **kw)
    Ar, Br, _ = _run(ref, csr, csc, A0, B0, method, numiter, **kw)
    assert np.isfinite(Ao).all() and np.isfinite(Bo).all()
    if _EXACT["on"]:
        assert np.array_equal(Ao, Ar) and np.array_equal(Bo, Br)
        return
    oo = harness.poisson_objective(Ao, Bo, csr, args["l2_reg"], args["l1_reg"], args["w_mult"])
    orf = harness.poisson_objective(Ar, Br, csr, args["l2_reg"], args["l1_reg"], args["w_mult"])
    if method == "pg":
        assert H.scaled_err(Ao, Ar) <= tol(is_float, 1e-12, 1e-5)
        assert H.scaled_err(Bo, Br) <= tol(is_float, 1e-12, 1e-5)
    elif method == "cg":
        if is_float and args["numiter"] < 10:
            assert abs(oo - orf) <= 2e-2 * abs(orf)   # fp32 mid-path: see test_g3_cg_half
        elif is_float:
            assert H.frac_rows_close(Ao, Ar, 5e-2) >= 0.9 and H.frac_rows_close(Bo, Br, 5e-2) >= 0.9
            assert abs(oo - orf) <= 1e-5 * abs(orf)
        else:
            assert H.scaled_err(Ao, Ar) <= 5e-3 and H.scaled_err(Bo, Br) <= 5e-3
            assert abs(oo - orf) <= 1e-8 * abs(orf)
    else:
        assert abs(oo - orf) <= tol(is_float, 1e-5, 1e-2) * abs(orf)


@pytest.mark.parametrize("method", ["pg", "cg", "tncg"])
@pytest.mark.parametrize("numiter", [1, 2, 3, "default"])
def test_g4_run_poismf_c1(libs, method, numiter):
    orc, ref, is_float = libs
    csr, csc, A0, B0 = H.c1_problem(is_float)
    _check_full(orc, ref, is_float, csr, csc, A0, B0, method, numiter)


@pytest.mark.parametrize("early_stop,reuse_prev", [(True, True), (False, True), (False, False)])
def test_g4_tncg_toggles(libs, early_stop, reuse_prev):
    orc, ref, is_float = libs
    csr, csc, A0, B0 = H.c1_problem(is_float)
    _check_full(orc, ref, is_float, csr, csc, A0, B0, "tncg", 3, early_stop=early_stop, reuse_prev=reuse_prev)


@pytest.mark.parametrize("method", ["pg", "cg", "tncg"])
def test_g5_edges(libs, method):
    """empty rows and columns, w_mult != 1, l1 > 0, power-law columns"""
    orc, ref, is_float = libs
    csr, csc, A0, B0 = H.small_problem(60, 90, 900, 8, is_float, seed=3, empty_rows=(0, 17, 59),
                                       empty_cols=(5, 89), powerlaw=True)
    _check_full(orc, ref, is_float, csr, csc, A0, B0, method, 2, k=8)
    _check_full(orc, ref, is_float, csr, csc, A0, B0, method, 2, k=8, w_mult=3.0, l1_reg=0.5)
    _check_full(orc, ref, is_float, csr, csc, A0, B0, method, 2, k=8, limit_step=False)


def test_threads_do_not_change_results(libs):
    orc, ref, is_float = libs
    csr, csc, A0, B0 = H.c1_problem(is_float)
    for method in ("pg", "cg"):
        a1, b1, _ = _run(orc, csr, csc, A0, B0, method, 2, nthreads=1)
        a4, b4, _ = _run(orc, csr, csc, A0, B0, method, 2, nthreads=4)
        assert np.array_equal(a1, a4) and np.array_equal(b1, b4)


# ---------------------------------------------------------------- N1: factors_multiple
@pytest.mark.parametrize("method", ["pg", "cg", "tncg"])
@pytest.mark.parametrize("w", [1.0, 3.0])
@pytest.mark.parametrize("reuse", [True, False])
def test_factors_multiple(libs, method, w, reuse):
    """ref: src/pred.c:66-199.  Bit-exact in the BLAS flavour; the portable flavour to the usual tolerances."""
    orc, ref, is_float = libs
    csr, csc, A0, B0 = H.small_problem(60, 90, 900, 8, is_float, seed=3, empty_rows=(0, 17), powerlaw=True)
    Bsum = (B0.astype(np.float64).sum(0) + 0.25).astype(B0.dtype)
    Amean = A0.mean(0).astype(B0.dtype)
    l2, maxupd, _ = harness.auto_defaults(method, 8)
    a = orc.factors_multiple(B0, Bsum, Amean, csr[0], csr[2], csr[1], l2, w, 1e-7, 3, maxupd, method, True, reuse)
    b = ref.factors_multiple(B0, Bsum, Amean, csr[0], csr[2], csr[1], l2, w, 1e-7, 3, maxupd, method, True, reuse)
    assert not a[[0, 17]].any() and not b[[0, 17]].any()
    if _EXACT["on"]:
        assert np.array_equal(a, b)
    elif method == "pg":
        assert H.scaled_err(a, b) <= tol(is_float, 1e-12, 1e-5)
    else:
        fo = H.half_objective(a, B0, csr[0], csr[1], csr[2], Bsum, l2 if method == "cg" else 0.0, w)
        fr = H.half_objective(b, B0, csr[0], csr[1], csr[2], Bsum, l2 if method == "cg" else 0.0, w)
        assert abs(fo - fr) <= tol(is_float, 1e-6, 2e-2) * abs(fr)


# ---------------------------------------------------------------- N4: predict_multiple, topN
def _serve_inputs(is_float, seed=0):
    rng = np.random.default_rng(seed)
    dt = np.float32 if is_float else np.float64
    A = rng.random((300, 20)).astype(dt)
    B = rng.random((5000, 20)).astype(dt)
    return rng, A, B


def test_predict_multiple(libs):
    orc, ref, is_float = libs
    rng, A, B = _serve_inputs(is_float)
    ia = rng.integers(0, 300, 2000).astype(np.uint64)
    ib = rng.integers(0, 5000, 2000).astype(np.uint64)
    o, r = orc.predict_multiple(A, B, ia, ib), ref.predict_multiple(A, B, ia, ib)
    if _EXACT["on"]:
        assert np.array_equal(o, r)
    else:
        assert H.scaled_err(o, r) <= tol(is_float, 1e-14, 1e-6)


@pytest.mark.parametrize("case", ["all", "include", "exclude_many", "exclude_few", "most"])
def test_topn(libs, case):
    orc, ref, is_float = libs
    rng, A, B = _serve_inputs(is_float, 1)
    none = np.empty(0, np.uint64)
    inc, exc, nt = {"all": (none, none, 10),
                    "include": (np.sort(rng.choice(5000, 300, replace=False)).astype(np.uint64), none, 10),
                    "exclude_many": (none, np.sort(rng.choice(5000, 1000, replace=False)).astype(np.uint64), 25),
                    "exclude_few": (none, rng.choice(5000, 40, replace=False).astype(np.uint64), 7),
                    "most": (none, none, 4000)}[case]
    a = A[3].copy()
    for lib in (orc, ref):
        rc, ix, sc = lib.topn(a, B, inc, exc, nt)
        assert rc == 0
        H.check_topn(a, B, ix, sc, inc, exc, nt, tol(is_float, 1e-13, 1e-5) if not _EXACT["on"] else (1e-5 if is_float else 1e-13))
    if case != "most":  # no near-ties among the first few: identical answers
        assert np.array_equal(orc.topn(a, B, inc, exc, nt)[1], ref.topn(a, B, inc, exc, nt)[1])
    assert orc.topn(a, B, inc[:3] if len(inc) else np.arange(3, dtype=np.uint64), np.arange(3, dtype=np.uint64), 2)[0] == 2
    assert ref.topn(a, B, np.arange(3, dtype=np.uint64), np.arange(3, dtype=np.uint64), 2)[0] == 2
