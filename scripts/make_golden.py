#!/usr/bin/env python3
"""Mint the golden vectors in tests/golden/ from the REAL reference compiled in place
(oracle/_ref/libpoismf_ref_{d,f}.so, built by `make -C oracle ref` from /root/reference/src).

Only runs where /root/reference exists (the development container).  The fixtures are data: inputs and
the reference's outputs.  Levels follow SURVEY.md section 8c (G1 primitives ... G5 edge cases).

    python scripts/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import bindings  # noqa: E402
from poismf_amd import harness  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def rows_fixture(ref, is_float):
    """G1 + G2: single-row primitives and solvers."""
    d = {}
    cases = [(5, 1, 1.0, 0.0), (5, 7, 10.0, 0.5), (50, 100, 1.0, 0.0), (100, 300, 1.0, 0.0), (50, 7, 10.0, 0.5)]
    for ci, (k, nnz, w, l1) in enumerate(cases):
        F, a, bsum, xval, xind = H.random_row(k, nnz, 600, is_float, seed=100 + ci, l1=l1)
        p = f"r{ci}_"
        d.update({p + "F": F, p + "a": a, p + "bsum": bsum, p + "xval": xval, p + "xind": xind,
                  p + "w": np.float64(w), p + "l2cg": np.float64(1e4), p + "l2tn": np.float64(1e3)})
        d[p + "grad_pgd"] = ref.calc_grad_pgd(a, F, xval, xind)
        d[p + "fun_single"] = np.float64(ref.calc_fun_single(a, F, bsum, xval, xind, 1e4, w))
        d[p + "grad_single"] = ref.calc_grad_single(a, F, bsum, xval, xind, 1e4, w, False)
        d[p + "grad_single_w"] = ref.calc_grad_single(a, F, bsum, xval, xind, 1e4, w, True)
        f, g = ref.calc_fun_and_grad(a, F, bsum, xval, xind, 1e3, w)
        d[p + "fg_f"], d[p + "fg_g"] = np.float64(f), g
        for limit_step in (True, False):
            for maxupd in (1, 5):
                x, f, ni, nf, rc = ref.cg_row(a, F, bsum, xval, xind, 1e4, w, maxupd, limit_step)
                q = p + f"cg_{int(limit_step)}_{maxupd}_"
                d[q + "x"], d[q + "meta"] = x, np.array([f, ni, nf, rc], np.float64)
        for reuse in (True, False):
            for maxupd in (10, 75, 750):
                a0 = a if reuse else np.full_like(a, 1e-3)
                x, f, nf, ni, rc = ref.tnc_row(a0, F, bsum, xval, xind, 1e3, w, maxupd)
                q = p + f"tnc_{int(reuse)}_{maxupd}_"
                d[q + "x"], d[q + "meta"] = x, np.array([f, nf, ni, rc], np.float64)
    d["ncases"] = np.int64(len(cases))
    return d


def run_full(ref, csr, csc, A0, B0, method, numiter, k, **kw):
    l2, maxupd, niter = harness.auto_defaults(method, k)
    args = dict(l2_reg=l2, l1_reg=0.0, w_mult=1.0, step_size=1e-7, method=method, limit_step=True,
                numiter=niter if numiter == "default" else numiter, maxupd=maxupd, early_stop=True,
                reuse_prev=False)
    args.update(kw)
    A, B = A0.copy(), B0.copy()
    rc = ref.run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], **args)
    assert rc == 0
    return A, B


def full_fixture(ref, is_float):
    """G3 + G4 on config C1 (the README data), G5 on a small matrix with empty rows/cols."""
    d = {}
    csr, csc, A0, B0 = H.c1_problem(is_float)
    d.update({"c1_csr_data": csr[0], "c1_csr_indices": csr[1].astype(np.int32), "c1_csr_indptr": csr[2].astype(np.int32),
              "c1_csc_data": csc[0], "c1_csc_indices": csc[1].astype(np.int32), "c1_csc_indptr": csc[2].astype(np.int32),
              "c1_A0": A0, "c1_B0": B0})
    # G3: one half-sweep of each kind, A side (CSR) against B0
    step, l2pg = 1e-7, 1e9
    cs = ref.sum_by_cols(B0)
    d["g3_colsum_B0"] = cs.copy()
    A = A0.copy()
    ref.pg_iteration(A, B0, csr[0], csr[2], csr[1], 1.0 / (1.0 + 2.0 * l2pg * step), cs * (-step), None, step, 1.0, 10)
    d["g3_pg_A"] = A
    for limit_step in (True, False):
        A = A0.copy()
        ref.cg_iteration(A, B0, csr[0], csr[2], csr[1], limit_step, cs, 1e4, 1.0, 5)
        d[f"g3_cg{int(limit_step)}_A"] = A
    for reuse in (True, False):
        A = A0.copy()
        conv = ref.tncg_iteration(A, B0, reuse, csr[0], csr[2], csr[1], cs, 1e3, 1.0, 75, True)
        d[f"g3_tncg{int(reuse)}_A"] = A
        d[f"g3_tncg{int(reuse)}_conv"] = np.int64(conv)
    # G4
    for method in ("pg", "cg", "tncg"):
        for numiter in (1, 2, 3, "default"):
            A, B = run_full(ref, csr, csc, A0, B0, method, numiter, 5)
            d[f"g4_{method}_{numiter}_A"], d[f"g4_{method}_{numiter}_B"] = A, B
    for early_stop, reuse_prev in ((True, True), (False, True), (False, False)):
        A, B = run_full(ref, csr, csc, A0, B0, "tncg", 3, 5, early_stop=early_stop, reuse_prev=reuse_prev)
        d[f"g4_tncg_es{int(early_stop)}_rp{int(reuse_prev)}_A"] = A
        d[f"g4_tncg_es{int(early_stop)}_rp{int(reuse_prev)}_B"] = B
    # G5
    csr, csc, A0, B0 = H.small_problem(60, 90, 900, 8, is_float, seed=3, empty_rows=(0, 17, 59),
                                       empty_cols=(5, 89), powerlaw=True)
    d.update({"g5_csr_data": csr[0], "g5_csr_indices": csr[1].astype(np.int32), "g5_csr_indptr": csr[2].astype(np.int32),
              "g5_csc_data": csc[0], "g5_csc_indices": csc[1].astype(np.int32), "g5_csc_indptr": csc[2].astype(np.int32),
              "g5_A0": A0, "g5_B0": B0})
    for method in ("pg", "cg", "tncg"):
        for tag, kw in (("plain", {}), ("w3_l1", dict(w_mult=3.0, l1_reg=0.5)), ("nolimit", dict(limit_step=False))):
            A, B = run_full(ref, csr, csc, A0, B0, method, 2, 8, **kw)
            d[f"g5_{method}_{tag}_A"], d[f"g5_{method}_{tag}_B"] = A, B
    return d


def factors_fixture(ref, is_float):
    """factors_multiple (ref src/pred.c:66-199) on the edge matrix, fitted-model inputs synthesised."""
    d = {}
    csr, csc, A0, B0 = H.small_problem(60, 90, 900, 8, is_float, seed=3, empty_rows=(0, 17, 59), empty_cols=(5, 89),
                                       powerlaw=True)
    Bsum = (B0.astype(np.float64).sum(0) + 0.25).astype(B0.dtype)
    Amean = A0.mean(0).astype(B0.dtype)
    d.update({"csr_data": csr[0], "csr_indices": csr[1].astype(np.int32), "csr_indptr": csr[2].astype(np.int32),
              "B": B0, "Bsum": Bsum, "Amean": Amean})
    for method in ("pg", "cg", "tncg"):
        l2, maxupd, _ = harness.auto_defaults(method, 8)
        for w in (1.0, 3.0):
            for reuse in (True, False):
                d[f"{method}_w{int(w)}_r{int(reuse)}"] = ref.factors_multiple(B0, Bsum, Amean, csr[0], csr[2], csr[1], l2, w,
                                                                             1e-7, 3, maxupd, method, True, reuse)
    return d


def main():
    bindings.build(ref=True)
    os.makedirs(OUT, exist_ok=True)
    for is_float in (False, True):
        ref = bindings.Reference(is_float)
        tag = "f32" if is_float else "f64"
        np.savez_compressed(os.path.join(OUT, f"rows_{tag}.npz"), **rows_fixture(ref, is_float))
        np.savez_compressed(os.path.join(OUT, f"full_{tag}.npz"), **full_fixture(ref, is_float))
        np.savez_compressed(os.path.join(OUT, f"factors_{tag}.npz"), **factors_fixture(ref, is_float))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
