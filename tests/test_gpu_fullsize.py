"""Parity at a BASELINE full size (config C2: uniform 1e5 x 1e5, 1e7 triplets, k = 50) through properties that
do not need the oracle to process the whole matrix:

* sampled-row parity: rows are independent given the opposing factor, so for a random sample of rows the
  oracle's half-sweep on the sub-matrix made of just those rows must reproduce the GPU's rows;
* non-negativity, empty rows exactly zero, finiteness;
* the column-sum vector the GPU used (recomputed on the host in fp64) is consistent with the result.
"""
import os

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


# ======================================================================================================================
# Configs C3 / C4 (one 1e6 x 1e5, 1e8-triplet matrix, k = 50) and C5 (Last.FM-shaped, k = 100) at FULL size.
# The oracle cannot process these matrices in test time, rows are independent given the opposing factor: for a
# random sample of rows of each half the oracle's half-sweep on the sub-matrix of just those rows must reproduce
# the GPU's rows (ref: src/poismf.c:139-188 pg_iteration, :275-322 cg_iteration, :324-404 tncg_iteration).
# Sessions are built on the device from the triplets (the CSR / CSC of 1e8 nonzeros never exist on the host).
# ======================================================================================================================
import scipy.sparse as sp  # noqa: E402

SAMPLE = 300


def _sub_from_triplets(trip, which, rows, use_float):
    """CSR (which = 1: rows of A) or CSC (which = 0: rows of B) of just `rows` (sorted), duplicates summed and
    indices sorted as the session's own conversion does; returns (data, indices, indptr) with local row ids."""
    major, minor = (trip.row, trip.col) if which else (trip.col, trip.row)
    dim_minor = trip.shape[1] if which else trip.shape[0]
    sel = np.flatnonzero(np.isin(major, rows))
    local = np.searchsorted(rows, major[sel])
    m = sp.csr_matrix((trip.data[sel], (local, minor[sel])), shape=(len(rows), dim_minor))
    m.sum_duplicates(); m.sort_indices()
    dt = np.float32 if use_float else np.float64
    return (np.ascontiguousarray(m.data, dtype=dt), np.ascontiguousarray(m.indices, dtype=np.uint64),
            np.ascontiguousarray(m.indptr, dtype=np.uint64))


def _row_objectives(M, F, data, indices, indptr, bsum, l2, w=1.0):
    """per row: bsum.a + l2 |a|^2 - w sum_j x_j log(a.F_j) in fp64"""
    M64, F64 = np.asarray(M, np.float64), np.asarray(F, np.float64)
    ip = indptr.astype(np.int64)
    rows = np.repeat(np.arange(M64.shape[0]), np.diff(ip))
    pred = np.einsum("ij,ij->i", M64[rows], F64[indices.astype(np.int64)])
    with np.errstate(all="ignore"):
        ll = np.bincount(rows, weights=np.asarray(data, np.float64) * np.log(pred), minlength=M64.shape[0])
    return M64 @ np.asarray(bsum, np.float64) + l2 * (M64 ** 2).sum(1) - w * ll


def _row_lengths(trip, which):
    """distinct nonzeros per row of the half (duplicates collapse): only emptiness is needed here"""
    major = trip.row if which else trip.col
    return np.bincount(major, minlength=trip.shape[0] if which else trip.shape[1])


def _stratified_rows(rng, lengths, plan, total):
    """Rows to check, taken from EVERY launch of the half-sweep just run (round 4; a uniform sample of a power-law half never met the
    rows that matter: config C5's 60 rows above 8192 nonzeros, config C3's 300 multi-CU rows).  `plan` lists the launches in the
    order of the device's row sort (longest rows first); position p of that order is approximated by sorting the triplet counts --
    duplicates move a few rows across a boundary, no launch is missed.  Every launch contributes min(its rows, total / launches,
    at least 24) rows -- all of its rows when it has fewer -- and a uniform sample fills up to `total`.
    Returns (sorted rows, {launch name: rows taken})."""
    order = np.argsort(-np.asarray(lengths, np.int64), kind="stable")
    per = max(24, total // max(len(plan), 1))
    picks, taken, pos = [], {}, 0
    for name, n in plan:
        seg = order[pos:pos + n]
        pos += n
        m = min(len(seg), per)
        if m:
            picks.append(rng.choice(seg, m, replace=False))
            taken[name] = taken.get(name, 0) + m
    got = np.unique(np.concatenate(picks)) if picks else np.zeros(0, np.int64)
    if len(got) < total:
        rest = np.setdiff1d(rng.choice(len(lengths), min(len(lengths), 2 * total), replace=False), got)[:total - len(got)]
        got = np.union1d(got, rest)
    return np.sort(got), taken


def _fullsize_halves(trip, k, method, use_float, maxupd, reuse_prev=True, warm_sweeps=0):
    """Runs one B half and one A half on the full matrix and checks SAMPLE rows of each against the oracle.  Returns the
    observed maxima so that the caller's tolerances are measured numbers.  warm_sweeps > 0: that many full sweeps run first, so
    that the checked halves are steady-state ones (rows near their optimum: fewer evaluations, other branches of the solvers)."""
    dimA, dimB = trip.shape
    A0, B0 = harness.initialize_matrices(dimA, dimB, k, use_float, 1)
    l2, mu, _ = harness.auto_defaults(method, k)
    maxupd = mu if maxupd is None else maxupd
    orc = H.checker(use_float, method)
    s = api.Session.from_coo(trip, k, use_float)
    stats = {}
    try:
        s.set_factors(A0, B0)
        if method == "cg":
            s.profile(True)        # the row kernels then record every row's (iterations, evaluations, rc)
        p = s.make_params(method, l2, maxupd=maxupd, reuse_prev=reuse_prev)
        step = s.real(1e-7)
        cnst_div = s.cnst_div(l2, step)
        rng = np.random.default_rng(7)
        prevA, prevB = A0, B0
        for _ in range(warm_sweeps):
            step = s.sweep(p, step)
            cnst_div = s.cnst_div(l2, step)
        if warm_sweeps:
            prevA, prevB = s.get_factors()
        for which in (0, 1):
            if which == 1 and method == "pg":
                step = s.real(step * 0.5)
            s.half_sweep(which, p, step, cnst_div)
            dec = s.decisions(which) if method == "cg" else None
            A1, B1 = s.get_factors()
            M1, F = (A1, prevB) if which else (B1, prevA)
            Mprev = prevA if which else prevB
            rows, strata = _stratified_rows(rng, _row_lengths(trip, which), s.plan(which), SAMPLE)
            stats[f"strata{which}"] = strata
            sd, si, sptr = _sub_from_triplets(trip, which, rows, use_float)
            G = M1[rows]
            # Two column-sum vectors.  "ref": the reference's serial real_t accumulation in row order (sum_by_cols, ref:
            # src/poismf.c:77-83) -- in fp32 over 1e6 rows that sum carries a relative error of ~4e-3 of its own (adding 0.3
            # to a partial sum of 3e5 rounds to multiples of 1/32).  "exact": the fp64 sum rounded once, which is what
            # the device's pairwise tree (4.2 in DESIGN.md) reproduces to an ulp.  The row kernels are judged given the
            # same Bsum ("exact", tight); the distance to the reference-with-its-own-sum is reported and bounded too.
            bs_ref = orc.sum_by_cols(F)
            bs_exact = F.astype(np.float64).sum(0).astype(F.dtype)
            for tag, bs in (("", bs_exact), ("_refsum", bs_ref)):
                Ms = np.ascontiguousarray(Mprev[rows])
                if method == "pg":
                    cs = bs * np.asarray(-step, bs.dtype)
                    if which:
                        cs = cs * np.asarray(-step, bs.dtype)   # quirk Q1
                    with np.errstate(all="ignore"):
                        orc.pg_iteration(Ms, F, sd, sptr, si, cnst_div, cs, None, step, 1.0, maxupd)
                    assert np.array_equal(np.isfinite(G), np.isfinite(Ms)) and np.array_equal(np.isnan(G), np.isnan(Ms))
                    fin = np.isfinite(Ms)
                    stats[f"err{which}{tag}"] = H.scaled_err(G[fin], Ms[fin]) if fin.any() else 0.0
                else:
                    if method == "cg":
                        orc.cg_iteration(Ms, F, sd, sptr, si, True, bs, l2, 1.0, maxupd)
                    else:
                        orc.tncg_iteration(Ms, F, reuse_prev, sd, sptr, si, bs, l2, 1.0, maxupd, False)
                    stats[f"err{which}{tag}"] = H.scaled_err(G, Ms)
                    l2o = l2 if method == "cg" else 0.0   # quirk Q4: TNC's objective has no l2 term
                    fo = _row_objectives(G, F, sd, si, sptr, bs, l2o)
                    fr = _row_objectives(Ms, F, sd, si, sptr, bs, l2o)
                    stats[f"obj{which}{tag}"] = abs(fo.sum() - fr.sum()) / abs(fr.sum())
                    stats[f"objrow{which}{tag}"] = float(np.max(np.abs(fo - fr) / np.maximum(np.abs(fr), 1e-300)))
                    stats[f"rows_close{which}{tag}"] = float(np.mean(np.abs(fo - fr) <= 1e-4 * np.abs(fr)))
                    if method == "tncg" and not tag:
                        # The reference against ITSELF on the same rows: the plain-loop build of the restatement against the build that
                        # sums through the reference's BLAS (bit-identical to the compiled reference) -- two summation orders of one
                        # algorithm.  From a cold start (reuse_prev off: every coordinate at 1e-3) or near a row's optimum TNC's line
                        # search fails or succeeds on rounding, in the reference too; this is the yardstick for such rows.
                        plain = bindings.Oracle(use_float)
                        Mp_ = np.ascontiguousarray(Mprev[rows])
                        plain.tncg_iteration(Mp_, F, reuse_prev, sd, sptr, si, bs, l2, 1.0, maxupd, False)
                        fp_ = _row_objectives(Mp_, F, sd, si, sptr, bs, l2o)
                        stats[f"self{which}"] = abs(fp_.sum() - fr.sum()) / abs(fr.sum())
                        stats[f"self_rows_close{which}"] = float(np.mean(np.abs(fp_ - fr) <= 1e-4 * np.abs(fr)))
                    if os.environ.get("POISMF_TEST_VERBOSE") and not tag:
                        lens = np.diff(sptr.astype(np.int64))
                        worst = np.argsort(-np.abs(fo - fr) / np.maximum(np.abs(fr), 1e-300))[:5]
                        for r in worst:
                            print(f"   half {which} row {rows[r]} nnz {lens[r]}: objective gpu {fo[r]:.12g} oracle {fr[r]:.12g}")
            if dec is not None:
                # the solver's decisions on the sampled rows against the checker's minimize_nonneg_cg (ref: src/nonnegcg.c:177-346)
                ip = sptr.astype(np.int64)
                same, dnf = 0, []
                for j, r in enumerate(rows):
                    xv, xi = np.ascontiguousarray(sd[ip[j]:ip[j + 1]]), np.ascontiguousarray(si[ip[j]:ip[j + 1]])
                    if len(xv) == 0:
                        same += 1
                        continue
                    _, _, ni_r, nf_r, rc_r = orc.cg_row(Mprev[r], F, bs_exact, xv, xi, l2, 1.0, maxupd, True)
                    same += (int(dec[0][r]), int(dec[1][r]), int(dec[2][r])) == (int(ni_r), int(nf_r), int(rc_r))
                    dnf.append(abs(int(dec[1][r]) - int(nf_r)))
                stats[f"dec_same{which}"] = same / len(rows)
                stats[f"dec_dnfeval{which}"] = float(np.mean(dnf)) if dnf else 0.0
            # invariants over the WHOLE factor
            empty = _row_lengths(trip, which) == 0
            assert not M1[empty].any()
            if not (method == "pg" and not np.isfinite(M1).all()):
                assert np.isfinite(M1).all() and (M1 >= 0).all()
            other1, other0 = (B1, prevB) if which else (A1, prevA)
            assert np.array_equal(other1, other0)       # the half that was not updated is untouched
            prevA, prevB = A1, B1
    finally:
        s.close()
    print(f"fullsize {method} {'f32' if use_float else 'f64'} k={k} maxupd={maxupd}: {stats}")
    return stats


@pytest.fixture(scope="module")
def c4_trip():
    return synth.uniform_triplets(10 ** 6, 10 ** 5, 10 ** 8, seed=1)


def test_c3_cg_fp64_fullsize(c4_trip):
    """BASELINE config C3: 1e6 x 1e5, 1e8 triplets, k = 50, cg fp64 (100-nonzero user rows, 1000-nonzero item rows)"""
    st = _fullsize_halves(c4_trip, 50, "cg", False, None)
    assert max(st["err0"], st["err1"]) <= 1e-3          # CG fp64 element-wise (measured 1.5e-10)
    assert max(st["obj0"], st["obj1"]) <= 1e-8          # measured 1e-14
    assert min(st["dec_same0"], st["dec_same1"]) >= 0.99   # the checker's (iterations, evaluations, rc) row for row


@pytest.mark.parametrize("maxupd", [None, 1])
def test_c4_pg_fp32_fullsize_on_one_gpu(c4_trip, maxupd):
    """BASELINE config C4's matrix and solver on one GPU (the 8-GPU run shards exactly these rows): Python default
    maxupd = 10 -- which overflows in the reference too (PG has no guard): same non-finite pattern -- and maxupd = 1"""
    st = _fullsize_halves(c4_trip, 50, "pg", True, maxupd)
    assert max(st["err0"], st["err1"]) <= 1e-5          # given the same column sums (measured 1.7e-7)
    assert max(st["err0_refsum"], st["err1_refsum"]) <= 2e-3   # against the reference's serial fp32 column sum (measured 4.2e-4)


def test_c4_cg_fp32_fullsize(c4_trip):
    """fp32 CG on 1000-nonzero rows is chaotic row by row in the reference too (whether a max_step-limited step leaves its
    coordinate at exactly 0 or at a 1e-9 residual decides the next iteration, oracle/poismf_oracle.c vaxpy): single rows
    end up to 20 % apart in objective, the sample's total within 4e-3 (B half, measured 3.9e-3) / 1e-5 (A half)."""
    st = _fullsize_halves(c4_trip, 50, "cg", True, None)
    assert st["obj0"] <= 1e-2 and st["obj1"] <= 1e-4
    # fp32 decisions: the same number of line-search trials within one per row on average (rows flip a backtracking step)
    assert max(st["dec_dnfeval0"], st["dec_dnfeval1"]) <= 1.5


@pytest.fixture(scope="module")
def c5_trip():
    c = synth.lastfm_like_coo()
    return synth.Triplets(c.row.astype(np.int64), c.col.astype(np.int64), np.asarray(c.data, np.float64), c.shape)


@pytest.mark.parametrize("reuse_prev,warm_sweeps", [(True, 0), (False, 0), (True, 2)])
def test_c5_tncg_fp64_fullsize(c5_trip, reuse_prev, warm_sweeps):
    """BASELINE config C5: Last.FM-shaped 358 858 x 160 112, ~17 M nnz, power-law item degrees (rows of > 1e5
    nonzeros: the long-row path), k = 100, tncg fp64, maxupd = 15 k.  Both `reuse_prev` settings (ref:
    src/poismf.c:379-381: rows restart at 1e-3 when it is off), and -- with reuse_prev -- the THIRD sweep as well as the first
    (steady state: rows start near their optimum, TNC stops on other criteria)."""
    st = _fullsize_halves(c5_trip, 100, "tncg", False, None, reuse_prev=reuse_prev, warm_sweeps=warm_sweeps)
    if reuse_prev and not warm_sweeps:
        assert max(st["obj0"], st["obj1"]) <= 1e-5      # TNCG fp64 objective (measured 4.7e-9 / 2.0e-10)
    else:
        # Cold starts (every coordinate at 1e-3) and third sweeps: some rows' line searches fail or succeed on rounding -- in the
        # reference as well: its plain-loop and its BLAS build end up to 20 % apart on single rows of this sample (measured: the
        # slot engine of round 2 and the lane engine of round 3 both show the same kind of rows, 3.7e-8 .. 1.1e-2 in the sample's
        # total).  The yardstick is the reference against itself on the SAME rows: the GPU must be as close to the compiled
        # reference's flavour as the other flavour is (x3), and agree with it on as many rows (-3 %).
        # Round 4: the samples are stratified by launch (150 of the A half's 300 rows now come from the 3 318 rows above 64 nonzeros, and
        # the B half's include all 60 rows above 8 192), i.e. they hold far more of the rows on which TNC's last accepted step is a
        # matter of rounding.  ONE such row ending 1.5 % (measured, A half) .. 29 % (B half, a 78-nonzero row; the reference's flavours
        # show 20 % on rows of the same kind) away moves a 300-row total by 5e-5 .. 1e-3, whichever implementation it happens to: the
        # floor of the total's bound is 2e-4 (was 5e-5 on uniform samples), the row-wise criterion is unchanged.
        for w in (0, 1):
            assert st[f"obj{w}"] <= max(2e-4, 3.0 * st[f"self{w}"]), (w, st[f"obj{w}"], st[f"self{w}"])
            assert st[f"rows_close{w}"] >= min(0.97, st[f"self_rows_close{w}"] - 0.03), (w, st[f"rows_close{w}"], st[f"self_rows_close{w}"])
