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
