"""Rows shared by a TEAM of CUs (poismf_amd/csrc/reg_eval.hpp, M_ > 1; poismf_hip.hip, half_sweep_team_kernel): CG on doubles
with two 16-byte slots per lane (k = 33 .. 64), rows of 385 .. 2048 nonzeros.  (Round 3: at k = 49 / 50 the lane-per-nonzero
engine, lane_eval.hpp, keeps rows of up to 1024 nonzeros in ONE CU, so teams see only longer rows there; the shapes below are
exercised at k = 48, the long ones at k = 50 too.)  Through
the C-ABI against the oracle: lengths on both sides of every team shape (2 x 32 steps up to 1024, 2 x 36 up to 1152, 3 x 28 up to
1344, 3 x 32 up to 1536, 4 x 32 up to 2048, streamed beyond), both line-search modes, the weighted objective (column sums of the tile cross
the team too), and agreement with the streamed path on the same input.  Needs an MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H
from tests.test_gpu_parity import compare, gpu_run, oracle_run
from tests.test_gpu_regtile import ragged_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# length classes are multiples of 64 up to 2048 nonzeros and every class maps to ONE team shape (decided by the class bound,
# never by the longest row that happens to share the bin: a row's summation order must not depend on its shard)
SHAPES = {
    "S=32,NW=4,M=2": [513, 700, 1000, 1023, 1024],
    "S=36,NW=4,M=2": [1025, 1100, 1151, 1152],
    "S=28,NW=4,M=3": [1153, 1200, 1343, 1344],
    "S=32,NW=4,M=3": [1345, 1500, 1535, 1536],
    "S=32,NW=4,M=4": [1537, 1800, 2047, 2048, 2049, 2300],
}
TEAM_LENGTHS = [384, 385, 386, 511, 512] + SHAPES["S=32,NW=4,M=2"] + SHAPES["S=32,NW=4,M=4"]


TEAM_K = 48   # 24 slots of 16 bytes: two per lane, not a lane-engine shape


def plan_of(csr, csc, A0, B0, k, which=1):
    """the launches a CG half-sweep of this problem takes"""
    from poismf_amd import api
    s = api.Session(csr, csc, A0.shape[0], B0.shape[0], k, False)
    s.set_factors(A0, B0)
    s.half_sweep(which, s.make_params("cg", 1e4), 1e-7, 1.0)
    txt = " ".join(name for name, _ in s.plan(which))
    s.close()
    return txt


NO_TEAMS = any(os.environ.get(k) for k in ("POISMF_HIP_NO_TEAM", "POISMF_HIP_NO_REGTILE", "POISMF_HIP_STATIC_ROWS"))   # scripts/knob_matrix.sh


@pytest.mark.parametrize("k", [TEAM_K, 50])
@pytest.mark.parametrize("shape", list(SHAPES))
def test_each_team_shape_vs_oracle(shape, k):
    if k == 50 and max(SHAPES[shape]) <= 1024:
        pytest.skip("k = 50 rows of up to 1024 nonzeros stay in one CU (lane engine)")
    csr, csc, A0, B0 = ragged_problem(SHAPES[shape], 6000, k, False, seed=5)
    if not NO_TEAMS:
        assert f"half_sweep_team_kernel<double,cg,{shape}>" in plan_of(csr, csc, A0, B0, k)
    A, B, args = gpu_run(csr, csc, A0, B0, "cg", 2, k)
    Ar, Br = oracle_run(False, csr, csc, A0, B0, "cg", args)
    compare(False, "cg", csr, args, A, B, Ar, Br, converged=False)


@pytest.mark.parametrize("k,limit_step,w_mult", [(TEAM_K, True, 1.0), (TEAM_K, False, 1.0), (TEAM_K, True, 2.5), (50, True, 1.0), (50, True, 2.5), (33, True, 1.0), (64, True, 1.0)])
def test_team_rows_vs_oracle(k, limit_step, w_mult):
    csr, csc, A0, B0 = ragged_problem(TEAM_LENGTHS, 6000, k, False, seed=5)
    A, B, args = gpu_run(csr, csc, A0, B0, "cg", 2, k, limit_step=limit_step, w_mult=w_mult)
    Ar, Br = oracle_run(False, csr, csc, A0, B0, "cg", args)
    assert not A[-1].any()   # the empty row
    compare(False, "cg", csr, args, A, B, Ar, Br, converged=False)


CHILD = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from tests.test_gpu_team import TEAM_LENGTHS
from tests.test_gpu_regtile import ragged_problem
from tests.test_gpu_parity import gpu_run
csr, csc, A0, B0 = ragged_problem(TEAM_LENGTHS, 6000, 48, False, seed=5)
A, B, _ = gpu_run(csr, csc, A0, B0, "cg", 2, 48)
np.save({out!r}, np.concatenate([A.ravel(), B.ravel()]))
"""


def test_team_and_streamed_paths_agree(tmp_path):
    """Same input with and without teams (the knob is read once per process, hence the children)."""
    res = {}
    for tag, env in (("team", {}), ("streamed", {"POISMF_HIP_NO_TEAM": "1"})):
        out = str(tmp_path / f"{tag}.npy")
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, out=out)], check=True, env=e, cwd=ROOT, timeout=600)
        res[tag] = np.load(out)
    assert np.isfinite(res["team"]).all()
    # two summation orders of the same fp64 arithmetic: mid-path CG rows may differ by one backtracking step (see compare())
    assert H.scaled_err(res["team"], res["streamed"]) <= 5e-3


def test_repeated_launches_are_deterministic():
    """Teams form in arrival order and rows come from a queue: which CUs hold a row changes from launch to launch, the bits
    of the result must not."""
    csr, csc, A0, B0 = ragged_problem(TEAM_LENGTHS * 3, 6000, TEAM_K, False, seed=9)
    first = None
    for _ in range(3):
        A, B, _ = gpu_run(csr, csc, A0, B0, "cg", 2, TEAM_K)
        if first is None:
            first = (A, B)
        else:
            assert np.array_equal(A, first[0]) and np.array_equal(B, first[1])


def test_segments_and_bin_company_do_not_change_team_rows():
    """fp64 CG rows of 1281 .. 2048 nonzeros (three different team shapes) give the same bits whether the half runs whole or cut
    into segments, i.e. whichever other rows share a row's bin: the team shape -- and with it each wave's share of the row and
    the summation order -- is a function of the row's length class alone (round-2 advisor finding: it used to follow the
    longest row of the bin)."""
    from poismf_amd import api, harness
    lengths = [1281, 1290, 1343, 1344, 1345, 1400, 1535, 1536, 1537, 1700, 2047, 2048] * 2 + [90, 100, 110, 600, 1000]
    csr, csc, A0, B0 = ragged_problem(lengths, 6000, 50, False, seed=21)
    dimA, dimB = A0.shape[0], B0.shape[0]
    l2, maxupd, _ = harness.auto_defaults("cg", 50)
    res = []
    for nseg in (1, 2, 5):
        s = api.Session(csr, csc, dimA, dimB, 50, False)
        s.set_factors(A0, B0)
        p = s.make_params("cg", l2, maxupd=maxupd)
        if nseg > 1:
            assert s.set_segments(1, nseg) == nseg
            for j in range(nseg):
                s.half_sweep(1, p, 1e-7, 1.0, seg=j)
        else:
            s.half_sweep(1, p, 1e-7, 1.0)
        res.append(s.get_factors()[0])
        s.close()
    assert np.isfinite(res[0]).all()
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])


def test_a_team_launch_that_gives_up_is_rerun_on_the_streamed_path(tmp_path):
    """POISMF_HIP_TEAM_SPIN_LIMIT=1 makes team leaders give up at once (their partner CU has not arrived within two polls): the
    launch sets the error word, the host puts the rows back where they started and runs them on the streamed LDS kernel.
    The call still returns 0, says so on stderr, and the factors equal those of POISMF_HIP_NO_TEAM=1 (the same kernel on the
    same rows from the same starting point)."""
    if any(os.environ.get(v) for v in ("POISMF_HIP_NO_TEAM", "POISMF_HIP_NO_REGTILE", "POISMF_HIP_STATIC_ROWS")):
        pytest.skip("no team launches under this knob (scripts/knob_matrix.sh)")
    res, err = {}, {}
    for tag, env in (("gave_up", {"POISMF_HIP_TEAM_SPIN_LIMIT": "1"}), ("streamed", {"POISMF_HIP_NO_TEAM": "1"})):
        out = str(tmp_path / f"{tag}.npy")
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, out=out)], check=True, env=e, cwd=ROOT, timeout=600,
                           capture_output=True, text=True)
        res[tag], err[tag] = np.load(out), r.stderr
    assert "re-run on the streamed path" in err["gave_up"], err["gave_up"]
    assert "out of memory" not in err["gave_up"]
    assert np.isfinite(res["gave_up"]).all()
    assert np.array_equal(res["gave_up"], res["streamed"])
