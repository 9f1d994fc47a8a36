"""The oracle against the committed golden vectors (tests/golden/*.npz, minted from the compiled
reference by scripts/make_golden.py).  Runs anywhere: needs neither /root/reference nor oracle/_ref.

Tolerances = the reference's own BLAS-to-BLAS variance; the bit-exact pin of the same
source against the reference lives in tests/test_oracle_vs_ref.py.
"""
import os

import numpy as np
import pytest

from oracle import bindings
from poismf_amd import harness
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", params=[False, True], ids=["f64", "f32"])
def ctx(request):
    is_float = request.param
    tag = "f32" if is_float else "f64"
    return (bindings.Oracle(is_float), is_float, np.load(os.path.join(GOLD, f"rows_{tag}.npz")),
            np.load(os.path.join(GOLD, f"full_{tag}.npz")))


def T(is_float, t64, t32):
    return t32 if is_float else t64


def test_g1_g2_rows(ctx):
    orc, is_float, rows, _ = ctx
    for ci in range(int(rows["ncases"])):
        p = f"r{ci}_"
        F, a, bsum, xval, xind = (rows[p + n] for n in ("F", "a", "bsum", "xval", "xind"))
        w = float(rows[p + "w"])
        t = T(is_float, 1e-12, 2e-5)
        assert H.scaled_err(orc.calc_grad_pgd(a, F, xval, xind), rows[p + "grad_pgd"]) <= t
        f = orc.calc_fun_single(a, F, bsum, xval, xind, 1e4, w)
        assert abs(f - float(rows[p + "fun_single"])) <= t * abs(float(rows[p + "fun_single"]))
        assert H.scaled_err(orc.calc_grad_single(a, F, bsum, xval, xind, 1e4, w, False), rows[p + "grad_single"]) <= t
        assert H.scaled_err(orc.calc_grad_single(a, F, bsum, xval, xind, 1e4, w, True), rows[p + "grad_single_w"]) <= t
        f, g = orc.calc_fun_and_grad(a, F, bsum, xval, xind, 1e3, w)
        assert abs(f - float(rows[p + "fg_f"])) <= t * abs(float(rows[p + "fg_f"]))
        assert H.scaled_err(g, rows[p + "fg_g"]) <= t
        for limit_step in (True, False):
            x, f, ni, nf, rc = orc.cg_row(a, F, bsum, xval, xind, 1e4, w, 1, limit_step)
            meta = rows[p + f"cg_{int(limit_step)}_1_meta"]
            assert (ni, rc) == (int(meta[1]), int(meta[3]))
            assert H.scaled_err(x, rows[p + f"cg_{int(limit_step)}_1_x"]) <= T(is_float, 1e-10, 1e-3)
            x, f, ni, nf, rc = orc.cg_row(a, F, bsum, xval, xind, 1e4, w, 5, limit_step)
            meta = rows[p + f"cg_{int(limit_step)}_5_meta"]
            assert abs(f - meta[0]) <= T(is_float, 1e-10, 2e-3) * abs(meta[0])
        for reuse in (True, False):
            for maxupd in (75, 750):
                a0 = a if reuse else np.full_like(a, 1e-3)
                x, f, nf, ni, rc = orc.tnc_row(a0, F, bsum, xval, xind, 1e3, w, maxupd)
                meta = rows[p + f"tnc_{int(reuse)}_{maxupd}_meta"]
                assert abs(f - meta[0]) <= T(is_float, 1e-6, 5e-2) * max(abs(meta[0]), 1.0)  # fp32 TNC is chaotic


def _mats(full, pre):
    u = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
    csr = (full[pre + "csr_data"], u(full[pre + "csr_indices"]), u(full[pre + "csr_indptr"]))
    csc = (full[pre + "csc_data"], u(full[pre + "csc_indices"]), u(full[pre + "csc_indptr"]))
    return csr, csc, full[pre + "A0"], full[pre + "B0"]


def test_fixture_inputs_match_generators(ctx):
    """the committed C1 inputs are exactly what the host harness builds from the README recipe"""
    _, is_float, _, full = ctx
    csr, csc, A0, B0 = _mats(full, "c1_")
    csr2, csc2, A02, B02 = H.c1_problem(is_float)
    for a, b in zip(csr + csc + (A0, B0), csr2 + csc2 + (A02, B02)):
        assert np.array_equal(a, b)
    assert len(csr[0]) == 9490  # 10 000 README triplets -> 9 490 after duplicate summing


def test_g3_half_sweeps(ctx):
    orc, is_float, _, full = ctx
    csr, csc, A0, B0 = _mats(full, "c1_")
    cs = orc.sum_by_cols(B0)
    assert H.scaled_err(cs, full["g3_colsum_B0"]) <= T(is_float, 1e-14, 1e-6)
    step, l2pg = 1e-7, 1e9
    A = A0.copy()
    orc.pg_iteration(A, B0, csr[0], csr[2], csr[1], 1.0 / (1.0 + 2.0 * l2pg * step), cs * (-step), None, step, 1.0, 10)
    assert H.scaled_err(A, full["g3_pg_A"]) <= T(is_float, 1e-12, 1e-5)
    if not is_float:
        for limit_step in (True, False):
            A = A0.copy()
            orc.cg_iteration(A, B0, csr[0], csr[2], csr[1], limit_step, cs, 1e4, 1.0, 5)
            assert H.scaled_err(A, full[f"g3_cg{int(limit_step)}_A"]) <= 1e-6
    for reuse in (True, False):
        A = A0.copy()
        conv = orc.tncg_iteration(A, B0, reuse, csr[0], csr[2], csr[1], cs, 1e3, 1.0, 75, True)
        assert conv == int(full[f"g3_tncg{int(reuse)}_conv"])
        fo = H.half_objective(A, B0, csr[0], csr[1], csr[2], cs, 1e3)
        fr = H.half_objective(full[f"g3_tncg{int(reuse)}_A"], B0, csr[0], csr[1], csr[2], cs, 1e3)
        assert abs(fo - fr) <= T(is_float, 1e-5, 1e-2) * abs(fr)


def _check(orc, is_float, csr, csc, A0, B0, Ar, Br, method, numiter, k, **kw):
    l2, maxupd, niter = harness.auto_defaults(method, k)
    args = dict(l2_reg=l2, l1_reg=0.0, w_mult=1.0, step_size=1e-7, method=method, limit_step=True,
                numiter=niter if numiter == "default" else numiter, maxupd=maxupd, early_stop=True,
                reuse_prev=False)
    args.update(kw)
    A, B = A0.copy(), B0.copy()
    assert orc.run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], **args) == 0
    oo = harness.poisson_objective(A, B, csr, args["l2_reg"], args["l1_reg"], args["w_mult"])
    orf = harness.poisson_objective(Ar, Br, csr, args["l2_reg"], args["l1_reg"], args["w_mult"])
    if method == "pg":
        assert H.scaled_err(A, Ar) <= T(is_float, 1e-12, 1e-5) and H.scaled_err(B, Br) <= T(is_float, 1e-12, 1e-5)
    elif method == "cg":
        if is_float and args["numiter"] < 10:
            assert abs(oo - orf) <= 2e-2 * abs(orf)
        elif is_float:
            assert abs(oo - orf) <= 1e-5 * abs(orf)
        else:
            assert H.scaled_err(A, Ar) <= 5e-3 and H.scaled_err(B, Br) <= 5e-3
            assert abs(oo - orf) <= 1e-8 * abs(orf)
    else:
        assert abs(oo - orf) <= T(is_float, 1e-5, 1e-2) * abs(orf)


@pytest.mark.parametrize("method", ["pg", "cg", "tncg"])
@pytest.mark.parametrize("numiter", [1, 2, 3, "default"])
def test_g4_full_c1(ctx, method, numiter):
    orc, is_float, _, full = ctx
    csr, csc, A0, B0 = _mats(full, "c1_")
    _check(orc, is_float, csr, csc, A0, B0, full[f"g4_{method}_{numiter}_A"], full[f"g4_{method}_{numiter}_B"],
           method, numiter, 5)


@pytest.mark.parametrize("early_stop,reuse_prev", [(True, True), (False, True), (False, False)])
def test_g4_tncg_toggles(ctx, early_stop, reuse_prev):
    orc, is_float, _, full = ctx
    csr, csc, A0, B0 = _mats(full, "c1_")
    tag = f"g4_tncg_es{int(early_stop)}_rp{int(reuse_prev)}_"
    _check(orc, is_float, csr, csc, A0, B0, full[tag + "A"], full[tag + "B"], "tncg", 3, 5,
           early_stop=early_stop, reuse_prev=reuse_prev)


@pytest.mark.parametrize("method", ["pg", "cg", "tncg"])
@pytest.mark.parametrize("tag,kw", [("plain", {}), ("w3_l1", dict(w_mult=3.0, l1_reg=0.5)), ("nolimit", dict(limit_step=False))])
def test_g5_edges(ctx, method, tag, kw):
    orc, is_float, _, full = ctx
    csr, csc, A0, B0 = _mats(full, "g5_")
    _check(orc, is_float, csr, csc, A0, B0, full[f"g5_{method}_{tag}_A"], full[f"g5_{method}_{tag}_B"], method, 2, 8, **kw)
    # quirk Q7: rows/columns without nonzeros are forced to exactly 0
    assert not full[f"g5_{method}_{tag}_A"][[0, 17, 59]].any() and not full[f"g5_{method}_{tag}_B"][[5, 89]].any()
