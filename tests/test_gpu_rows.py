"""The GPU's inner solvers pinned at the single-row level (golden levels G1 / G2) through the exported
C-ABI: `factors_multiple` (ref: src/pred.c:66-199) takes the column-sum vector and the starting point from the
caller, so a one-row CSR through it reaches the device's calc_grad_pgd (ref: src/poismf.c:126-133),
minimize_nonneg_cg (ref: src/nonnegcg.c:177-346) and tnc (ref: src/tnc.c:251-463) with exactly the inputs of
tests/golden/rows_f{32,64}.npz -- whose outputs were minted from the compiled reference (scripts/make_golden.py).

The observed maxima are printed (pytest -s) so that every tolerance below is a measured number with a margin, not a
constant picked to pass."""
import os

import numpy as np
import pytest

from poismf_amd import api
from tests import helpers as H

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", params=[False, True], ids=["f64", "f32"])
def rows(request):
    return request.param, np.load(os.path.join(GOLD, f"rows_{'f32' if request.param else 'f64'}.npz"))


def _case(z, ci):
    p = f"r{ci}_"
    F, a, bsum, xval, xind = (z[p + n] for n in ("F", "a", "bsum", "xval", "xind"))
    return p, F, a, bsum, xval, np.ascontiguousarray(xind, dtype=np.uint64), float(z[p + "w"])


def _one_row(F, bsum, start, xval, xind, w, **kw):
    """factors_multiple on a one-row CSR.  For w != 1 the entry point builds the row's constant term as
    (w - 1) sum_j F_j + Bsum (adjustment_Bsum, ref: src/poismf.c:85-123); the fixtures hand `bsum` to the solver as that
    term directly, so Bsum is chosen such that the sum comes out as the fixture's vector (to one rounding)."""
    dt = F.dtype
    if w != 1.0:
        bsum = (bsum.astype(np.float64) - (w - 1.0) * F[xind.astype(np.int64)].astype(np.float64).sum(0)).astype(dt)
    indptr = np.array([0, len(xval)], dtype=np.uint64)
    return api._predict_factors_multiple(np.ascontiguousarray(F), np.ascontiguousarray(bsum), np.ascontiguousarray(start), indptr, xind,
                                         np.ascontiguousarray(xval), w_mult=w, **kw)[0]


def _objective(x, F, bsum, xval, xind, l2, w):
    """bsum.x + l2 |x|^2 - w sum_j x_j log(x . F_j) in fp64 (calc_fun_single, ref: src/poismf.c:194-208; l2 = 0 gives the
    value TNC minimises, quirk Q4)"""
    x64, F64 = x.astype(np.float64), F.astype(np.float64)
    pred = F64[xind.astype(np.int64)] @ x64
    with np.errstate(all="ignore"):
        return float(bsum.astype(np.float64) @ x64 + l2 * (x64 @ x64) - w * (xval.astype(np.float64) * np.log(pred)).sum())


def test_g1_calc_grad_pgd_through_one_pg_update(rows):
    """One PG update with l2 = 0 (divisor 1) is  a' = max(0, (a + step g(a)) + (-step) Bsum): with the fixture's
    calc_grad_pgd output g the host repeats those three roundings in the working precision -- the device's gradient is
    pinned to the reference's to the precision with which a' resolves it."""
    use_float, z = rows
    dt = np.float32 if use_float else np.float64
    worst = 0.0
    for ci in range(int(z["ncases"])):
        p, F, a, bsum, xval, xind, w = _case(z, ci)
        if w != 1.0:
            continue   # the weighted PG path scales a different vector (Bsum_w); G5 covers it end to end
        g = z[p + "grad_pgd"].astype(dt)
        step = dt(0.05 * float(np.abs(a).max()) / max(float(np.abs(g).max()), 1e-30))   # the gradient term moves a by ~5 %
        got = _one_row(F, bsum, a, xval, xind, w, l2_reg=0.0, step_size=float(step), niter=1, maxupd=1, method="pg",
                       limit_step=False, reuse_mean=True)
        want = a + step * g
        want = want + bsum * dt(-step)
        want = np.maximum(want * dt(1.0), dt(0))
        # what the gradient itself is resolved to: error of a' divided by the size of the gradient term
        err = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64)))) / float(step * np.abs(g).max())
        worst = max(worst, err)
    print(f"G1 calc_grad_pgd {'f32' if use_float else 'f64'}: max error relative to max|step g| = {worst:.3g}")
    assert worst <= (2e-5 if use_float else 1e-10)


@pytest.mark.parametrize("limit_step", [True, False])
def test_g2_cg_rows(rows, limit_step):
    use_float, z = rows
    worst1, worst5, worst1_rel = 0.0, 0.0, 0.0
    for ci in range(int(z["ncases"])):
        p, F, a, bsum, xval, xind, w = _case(z, ci)
        l2 = float(z[p + "l2cg"])
        for maxiter in (1, 5):
            x = _one_row(F, bsum, a, xval, xind, w, l2_reg=l2, step_size=1e-7, niter=1, maxupd=maxiter, method="cg",
                         limit_step=limit_step, reuse_mean=True)
            want = z[p + f"cg_{int(limit_step)}_{maxiter}_x"]
            meta = z[p + f"cg_{int(limit_step)}_{maxiter}_meta"]
            if maxiter == 1:
                # One CG iteration is x1 = x0 + alpha d with alpha = max_step = min_i(-x_i / d_i) when the limited step is
                # accepted: the leading coordinate lands on 0 and the others are small DIFFERENCES of O(|x0|) numbers, so
                # the rounding of the gradient shows up relative to |x0|, not to the (possibly tiny) |x1|.
                scale = max(float(np.abs(a).max()), float(np.abs(want).max()))
                worst1 = max(worst1, float(np.abs(x.astype(np.float64) - want.astype(np.float64)).max()) / scale)
                worst1_rel = max(worst1_rel, H.scaled_err(x, want))
            else:
                f = _objective(x, F, bsum, xval, xind, l2, w)
                worst5 = max(worst5, abs(f - float(meta[0])) / abs(float(meta[0])))
    tag = 'f32' if use_float else 'f64'
    print(f"G2 cg limit_step={limit_step} {tag}: maxiter=1 element-wise {worst1:.3g} of max(|x0|, |x1|) ({worst1_rel:.3g} of max|x1|); "
          f"maxiter=5 objective {worst5:.3g}")
    assert worst1 <= (1e-5 if use_float else 1e-10)     # G2 with maxiter = 1
    assert worst5 <= (2e-3 if use_float else 1e-10)     # the bound tests/test_golden.py holds the oracle to


@pytest.mark.parametrize("reuse", [True, False])
def test_g2_tnc_rows(rows, reuse):
    use_float, z = rows
    worst, excess = {}, 0.0
    for ci in range(int(z["ncases"])):
        p, F, a, bsum, xval, xind, w = _case(z, ci)
        l2 = float(z[p + "l2tn"])
        for maxnfeval in (10, 75, 750):
            x = _one_row(F, bsum, a, xval, xind, w, l2_reg=l2, step_size=1e-7, niter=1, maxupd=maxnfeval, method="tncg",
                         limit_step=False, reuse_mean=reuse)
            meta = z[p + f"tnc_{int(reuse)}_{maxnfeval}_meta"]
            f = _objective(x, F, bsum, xval, xind, 0.0, w)          # quirk Q4: no l2 term in TNC's objective
            worst[maxnfeval] = max(worst.get(maxnfeval, 0.0), abs(f - float(meta[0])) / max(abs(float(meta[0])), 1.0))
            excess = max(excess, (f - float(meta[0])) / max(abs(float(meta[0])), 1.0))
    print(f"G2 tnc reuse={reuse} {'f32' if use_float else 'f64'}: objective rel diff by maxnfeval {worst}; worse than the reference by at most {excess:.3g}")
    # fp64: every budget agrees with the compiled reference to 1e-6 (measured: <= 6e-7).
    # fp32: TNC's finite-difference Hessian products make the path chaotic (the oracle and the compiled reference -- same
    # algorithm, different summation order -- end 1.6 % apart on case 3, scripts/probes/probe_rows.py): what is pinned
    # is (i) the GPU never ends materially ABOVE the reference's objective, at any budget, and (ii) once the budget is not
    # what stops the run (75, 750 evaluations) it lands within 1e-2 of it (measured <= 3e-3).  After only 10
    # evaluations from the 1e-3 start the reference itself is anywhere between 383 and 437 on case 2.
    # (fp32: 2e-3 held for the slot layout of rounds 1-2; the lane layout of round 3 -- another summation order again -- ends case 3
    # 3.2e-3 above the compiled reference at 75 / 750 evaluations, where the plain-loop and the BLAS build of the reference
    # itself are 1.6e-2 apart)
    assert excess <= (5e-3 if use_float else 1e-6)
    for mf, v in worst.items():
        if not use_float:
            assert v <= 1e-6
        elif mf >= 75:
            assert v <= 1e-2


# ---- G1: the objective / gradient wrappers themselves (poismf_hip_debug_row_eval) -----------------------------------------------------
def _row_eval(F, bsum, point, xval, xind, w, l2, which):
    """One row through the device's fun_single + grad_single (which = 0) or fun_and_grad (which = 1); `bsum` is the row's constant
    term as the fixtures hand it to the reference (see _one_row for w != 1)."""
    dt = F.dtype
    if w != 1.0:
        bsum = (bsum.astype(np.float64) - (w - 1.0) * F[xind.astype(np.int64)].astype(np.float64).sum(0)).astype(dt)
    indptr = np.array([0, len(xval)], dtype=np.uint64)
    f, G = api.debug_row_eval(np.ascontiguousarray(F), np.ascontiguousarray(bsum), np.ascontiguousarray(point), indptr, xind,
                              np.ascontiguousarray(xval), l2, w, which)
    return float(f[0]), G[0]


def test_g1_fun_single_grad_single_fun_and_grad_vs_the_compiled_reference(rows):
    """The golden rows' fun_single / grad_single / grad_single_w / fg_f / fg_g entries (minted from the compiled reference's
    calc_fun_single, calc_grad_single[_w] and calc_fun_and_grad, ref: src/poismf.c:194-273) against the DEVICE's own wrappers of the
    same names (solvers.hpp), evaluated by the row engine a CG half-sweep would pick for the row: k = 5 / 50 / 100, 1 .. 300
    nonzeros, w = 1 and 10.  With w != 1 the device takes the weighted gradient (quirk Q11), with w = 1 the plain one."""
    use_float, z = rows
    tol_f, tol_g = (2e-6, 2e-5) if use_float else (1e-13, 1e-12)
    worst_f = worst_g = 0.0
    for ci in range(int(z["ncases"])):
        p, F, a, bsum, xval, xind, w = _case(z, ci)
        f0, g0 = _row_eval(F, bsum, a, xval, xind, w, float(z[p + "l2cg"]), 0)
        f1, g1 = _row_eval(F, bsum, a, xval, xind, w, float(z[p + "l2tn"]), 1)
        g_ref = z[p + ("grad_single_w" if w != 1.0 else "grad_single")].astype(np.float64)
        for got, ref in ((f0, float(z[p + "fun_single"])), (f1, float(z[p + "fg_f"]))):
            worst_f = max(worst_f, abs(got - ref) / max(abs(ref), 1e-300))
        for got, ref in ((g0, g_ref), (g1, z[p + "fg_g"].astype(np.float64))):
            worst_g = max(worst_g, float(np.max(np.abs(got.astype(np.float64) - ref)) / max(float(np.max(np.abs(ref))), 1e-300)))
    print(f"G1 fun / grad wrappers {'f32' if use_float else 'f64'}: worst relative error of f {worst_f:.3g}, of g (scaled by max|g|) {worst_g:.3g}")
    assert worst_f <= tol_f and worst_g <= tol_g


@pytest.mark.parametrize("nnz", [40, 100, 200, 500, 1000, 1040, 1500, 3000])
def test_g1_wrappers_on_every_engine_vs_checker(rows, nnz):
    """The same wrappers on seeded k = 50 rows whose lengths walk through the engines of a CG half-sweep (register / lane instances with
    one and several waves, the partial LDS set, teams of CUs, streamed tiles) against the checker's calc_* functions."""
    use_float, _ = rows
    F, a, bsum, xval, xind = H.random_row(50, nnz, 4000, use_float, seed=7 + nnz)
    xind = np.ascontiguousarray(xind, dtype=np.uint64)
    orc = H.checker(use_float, "cg")
    tol_f, tol_g = (5e-6, 5e-5) if use_float else (1e-12, 1e-11)
    for w in (1.0, 3.0):
        f0, g0 = _row_eval(F, bsum, a, xval, xind, w, 1e4, 0)
        f1, g1 = _row_eval(F, bsum, a, xval, xind, w, 1e3, 1)
        rf0 = orc.calc_fun_single(a, F, bsum, xval, xind, 1e4, w)
        rg0 = orc.calc_grad_single(a, F, bsum, xval, xind, 1e4, w, w != 1.0)
        rf1, rg1 = orc.calc_fun_and_grad(a, F, bsum, xval, xind, 1e3, w)
        assert abs(f0 - rf0) <= tol_f * abs(rf0) and abs(f1 - rf1) <= tol_f * abs(rf1), (w, f0, rf0, f1, rf1)
        for got, ref in ((g0, rg0), (g1, rg1)):
            assert np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64))) <= tol_g * np.max(np.abs(ref)), w
