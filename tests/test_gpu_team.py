"""Rows shared by a TEAM of CUs (poismf_amd/csrc/reg_eval.hpp, M_ > 1; poismf_hip.hip, half_sweep_team_kernel): CG on doubles
with two 16-byte slots per lane (k = 33 .. 64), rows of 385 .. 2048 nonzeros -- the item rows of BASELINE config C3.  Through
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

# rows above 1024 nonzeros share one length class (powers of two there), and a class takes the shape its longest row needs
SHAPES = {
    "S=32,NW=4,M=2": [384, 385, 386, 511, 512, 513, 700, 1000, 1023, 1024],
    "S=36,NW=4,M=2": [1025, 1100, 1151, 1152],
    "S=28,NW=4,M=3": [1025, 1153, 1200, 1343, 1344],
    "S=32,NW=4,M=3": [1025, 1345, 1500, 1535, 1536],
    "S=32,NW=4,M=4": [1025, 1537, 1800, 2047, 2048, 2049, 2300],
}
TEAM_LENGTHS = SHAPES["S=32,NW=4,M=2"] + SHAPES["S=32,NW=4,M=4"]


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


@pytest.mark.parametrize("shape", list(SHAPES))
def test_each_team_shape_vs_oracle(shape):
    csr, csc, A0, B0 = ragged_problem(SHAPES[shape], 6000, 50, False, seed=5)
    if not NO_TEAMS:
        assert f"half_sweep_team_kernel<double,cg,{shape}>" in plan_of(csr, csc, A0, B0, 50)
    A, B, args = gpu_run(csr, csc, A0, B0, "cg", 2, 50)
    Ar, Br = oracle_run(False, csr, csc, A0, B0, "cg", args)
    compare(False, "cg", csr, args, A, B, Ar, Br, converged=False)


@pytest.mark.parametrize("k,limit_step,w_mult", [(50, True, 1.0), (50, False, 1.0), (50, True, 2.5), (33, True, 1.0), (64, True, 1.0)])
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
csr, csc, A0, B0 = ragged_problem(TEAM_LENGTHS, 6000, 50, False, seed=5)
A, B, _ = gpu_run(csr, csc, A0, B0, "cg", 2, 50)
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
    csr, csc, A0, B0 = ragged_problem(TEAM_LENGTHS * 3, 6000, 50, False, seed=9)
    first = None
    for _ in range(3):
        A, B, _ = gpu_run(csr, csc, A0, B0, "cg", 2, 50)
        if first is None:
            first = (A, B)
        else:
            assert np.array_equal(A, first[0]) and np.array_equal(B, first[1])
