"""bench.py's multi-GPU planning on CPU: the per-rank row ranges tile the matrix, are identical on every rank (they are
computed from the seeded triplets alone), balance the nonzeros, and the segments the sharded driver exchanges tile each
rank's range exactly as the C session cuts them (poismf_hip_host.hip, finish_half)."""
import importlib.util
import os

import numpy as np

from poismf_amd import dist as pdist
from poismf_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_plan_ranges_tile_and_balance():
    b = _bench()
    trip = synth.uniform_triplets(3000, 400, 60000, seed=1)
    for world in (1, 2, 4, 8):
        rA, rB = b.plan_ranges(trip, world)
        for r, dim, idx in ((rA, 3000, trip.row), (rB, 400, trip.col)):
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == dim
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            counts = np.bincount(idx, minlength=dim)
            loads = [int(counts[lo:hi].sum()) for lo, hi in r]
            assert max(loads) <= 1.05 * len(idx) / world + counts.max()
        assert (rA, rB) == b.plan_ranges(synth.uniform_triplets(3000, 400, 60000, seed=1), world)


def test_choose_ranges_falls_back_to_nnz_balance_on_skewed_rows():
    counts = np.r_[np.full(10, 1000), np.full(990, 1)]
    eq = pdist.equal_ranges(1000, 4)
    got = pdist.choose_ranges(counts, 4)
    assert got != eq and got[0][1] < 10          # the heavy rows are split over ranks
    assert pdist.choose_ranges(np.full(1000, 7), 4) == eq


def test_segments_tile_a_range_like_the_session_does():
    for rng in ((0, 10), (7, 1000), (5, 5), (125000, 250000)):
        for nseg in (1, 2, 3, 4, 7):
            segs = [pdist.segment_of(rng, j, nseg) for j in range(nseg)]
            assert segs[0][0] == rng[0] and segs[-1][1] == rng[1]
            assert all(segs[j][1] == segs[j + 1][0] for j in range(nseg - 1))
            n = rng[1] - rng[0]
            assert segs == [(rng[0] + n * j // nseg, rng[0] + n * (j + 1) // nseg) for j in range(nseg)]   # finish_half's cut


def test_sharded_generation_reproduces_the_one_shot_triplets():
    """bench.py --gpus N: a rank draws the row / column COUNTS of the whole matrix chunk by chunk (ranges are cut at their quantiles) and
    then only the triplets of its own row ranges (synth.uniform_counts / uniform_triplets_of) -- the same values, in the same order,
    as filtering the one-shot draw every rank used to make and hold."""
    dimA, dimB, n = 1000, 300, 200000
    t = synth.uniform_triplets(dimA, dimB, n, seed=1)
    cA, cB, st = synth.uniform_counts(dimA, dimB, n, seed=1, chunk=7777)
    assert np.array_equal(cA, np.bincount(t.row, minlength=dimA)) and np.array_equal(cB, np.bincount(t.col, minlength=dimB))
    rA, rB = pdist.choose_ranges(cA, 3), pdist.choose_ranges(cB, 3)
    assert (rA, rB) == _bench().plan_ranges(t, 3)
    seen = np.zeros(n, bool)
    for r in range(3):
        u = synth.uniform_triplets_of(dimA, dimB, n, rA[r], rB[r], st, seed=1, chunk=5000)
        m = ((t.row >= rA[r][0]) & (t.row < rA[r][1])) | ((t.col >= rB[r][0]) & (t.col < rB[r][1]))
        assert np.array_equal(u.row, t.row[m]) and np.array_equal(u.col, t.col[m]) and np.array_equal(u.data, t.data[m])
        seen |= m
    assert seen.all()


def test_traffic_is_quoted_only_for_this_trees_kernels(tmp_path, monkeypatch):
    """roofline.traffic comes from profiles/hbm_traffic.json (separate rocprofv3 --pmc passes, scripts/profile_round.sh); bench.py quotes
    it only while the file carries the hash of THIS tree's kernel sources + flags, and says where the number came from."""
    import json
    b = _bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    good = b.build._source_hash()
    (prof / "hbm_traffic.json").write_text(json.dumps({"round": "rXX", "source_hash": good, "C4_pg_maxupd10_f32": 5.0e10}))
    v, src = b.traffic_from_profiles("C4_pg_maxupd10_f32")
    assert v == 5.0e10 and "rXX" in src and good[:12] in src
    assert b.traffic_from_profiles("C4_cg_maxupd5_f64")[0] is None            # no such entry
    (prof / "hbm_traffic.json").write_text(json.dumps({"round": "rXX", "source_hash": "0" * 64, "C4_pg_maxupd10_f32": 5.0e10}))
    v, src = b.traffic_from_profiles("C4_pg_maxupd10_f32")
    assert v is None and "not quoted" in src


def test_valu_roofline_counts_the_references_flops():
    """roofline.valu: a PG pass is one gradient (4k+1 flops per nonzero), CG one gradient per iteration and one function value
    (2k+25) per trial, TNC 4k+26 per evaluation, against the vector peak of the precision."""
    b = _bench()

    class J:
        k, use_float = 50, True
    res = dict(method="pg", psteps=2, ev_stats=[(0, 1000), (0, 3000)], dec_stats=None)
    v = b.valu_block(J, res, sweep_ms=1.0)
    assert v["flops_per_sweep"] == (1000 + 3000) / 2 * 201 and v["peak"] == 157.3
    assert abs(v["achieved"] - v["flops_per_sweep"] / 1e-3 / 1e12) < 1e-9 and abs(v["frac"] - v["achieved"] / 157.3) < 1e-12
    J.use_float = False
    d = [dict(iterations=5, evaluations=9, nnz_iterations=500, nnz_evaluations=900)] * 2
    v = b.valu_block(J, dict(method="cg", psteps=1, ev_stats=None, dec_stats=d), sweep_ms=2.0)
    assert v["flops_per_sweep"] == 1000 * 201 + (1000 + 1800) * 125 and v["peak"] == 78.6
    v = b.valu_block(J, dict(method="tncg", psteps=1, ev_stats=None, dec_stats=d), sweep_ms=2.0)
    assert v["flops_per_sweep"] == 1800 * 226
