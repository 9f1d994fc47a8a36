"""Parity at a BASELINE full size (config C2: uniform 1e5 x 1e5, 1e7 triplets, k = 50) through properties that
do not need the oracle to process the whole matrix:

* sampled-row parity: rows are independent given the opposing factor, so for a random sample of rows the
  oracle's half-sweep on the sub-matrix made of just those rows must reproduce the GPU's rows;
* non-negativity, empty rows exactly zero, finiteness;
* the column-sum vector the GPU used (recomputed on the host in fp64) is consistent with the result.
"""
import numpy as np
import pytest

from oracle import bindings
from poismf_amd import api, harness, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

DIMA = DIMB = 10 ** 5
K = 50


@pytest.fixture(scope="module")
def c2_coo():
    return synth.uniform_coo(DIMA, DIMB, 10 ** 7, seed=1)


def _sub_csr(data, indices, indptr, rows):
    ip = indptr.astype(np.int64)
    lens = ip[rows + 1] - ip[rows]
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    idx = np.concatenate([np.arange(ip[r], ip[r + 1]) for r in rows])
    return np.ascontiguousarray(data[idx]), np.ascontiguousarray(indices[idx]), ptr


@pytest.mark.parametrize("method,use_float,maxupd_override", [("pg", True, 1), ("pg", True, None), ("cg", False, None)])
def test_c2_sampled_rows_vs_oracle(c2_coo, method, use_float, maxupd_override):
    """pg/maxupd=1 is the R default and the bandwidth point; pg with the Python default maxupd=10 drives rows to
    exactly zero on this matrix (the Bsum term dominates), divides by zero on the next inner update and fills
    them with inf -- in the reference as well (PG has no guard, ref poismf/__init__.py:37-41): there parity
    means the same non-finite pattern."""
    csr, csc = harness.process_data(c2_coo, use_float)
    assert len(csr[0]) == 9994947  # C2 after duplicate summing
    A0, B0 = harness.initialize_matrices(DIMA, DIMB, K, use_float, 1)
    l2, maxupd, _ = harness.auto_defaults(method, K)
    if maxupd_override is not None:
        maxupd = maxupd_override
    orc = bindings.Oracle(use_float)
    s = api.Session(csr, csc, DIMA, DIMB, K, use_float)
    s.set_factors(A0, B0)
    p = s.make_params(method, l2, maxupd=maxupd)
    step = 1e-7
    cnst_div = 1. / (1. + 2. * l2 * step)
    rng = np.random.default_rng(7)
    prevA, prevB = A0, B0
    for which in (0, 1):                      # B half, then A half
        if which == 1 and method == "pg":
            step *= 0.5
        s.half_sweep(which, p, step, cnst_div)
        A1, B1 = s.get_factors()
        M1, F = (A1, prevB) if which else (B1, prevA)
        Mprev = prevA if which else prevB
        data, indices, indptr = csr if which else csc
        rows = np.sort(rng.choice(M1.shape[0], 300, replace=False))
        sd, si, sp = _sub_csr(data, indices, indptr, rows)
        Ms = np.ascontiguousarray(Mprev[rows])
        bs = orc.sum_by_cols(F)
        if method == "pg":
            cs = bs * np.asarray(-step, bs.dtype)
            if which:
                cs = cs * np.asarray(-step, bs.dtype)   # quirk Q1
            with np.errstate(all="ignore"):
                orc.pg_iteration(Ms, F, sd, sp, si, cnst_div, cs, None, step, 1.0, maxupd)
            if not np.isfinite(Ms).all():
                assert np.array_equal(np.isfinite(M1[rows]), np.isfinite(Ms)) and np.array_equal(np.isnan(M1[rows]), np.isnan(Ms))
                fin = np.isfinite(Ms)
                assert np.allclose(M1[rows][fin], Ms[fin], rtol=1e-4, atol=0)
                prevA, prevB = A1, B1
                continue
            assert H.scaled_err(M1[rows], Ms) <= (1e-5 if use_float else 1e-12)
        else:
            orc.cg_iteration(Ms, F, sd, sp, si, True, bs, l2, 1.0, maxupd)
            assert H.scaled_err(M1[rows], Ms) <= (5e-2 if use_float else 5e-3)
            fo = H.half_objective(M1[rows], F, sd, si, sp, bs, l2)
            fr = H.half_objective(Ms, F, sd, si, sp, bs, l2)
            assert abs(fo - fr) <= 1e-8 * abs(fr)
        # invariants over the WHOLE factor
        if method == "pg" and maxupd_override is None:
            pass   # degenerate by construction (0 / inf rows), compared entry-wise above
        else:
            assert np.isfinite(M1).all() and (M1 >= 0).all()
        if not (method == "pg" and maxupd_override is None):
            assert M1[~(np.diff(indptr.astype(np.int64)) == 0)].any(axis=1).all()   # rows with data did not collapse
        empty = np.diff(indptr.astype(np.int64)) == 0
        assert not M1[empty].any()
        # the half that was not updated is untouched
        other1, other0 = (B1, prevB) if which else (A1, prevA)
        assert np.array_equal(other1, other0)
        prevA, prevB = A1, B1
    s.close()
