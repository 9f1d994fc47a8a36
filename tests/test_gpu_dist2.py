"""The sharded driver (poismf_amd/dist.py) with TWO REAL PROCESSES on the one GPU of the test box: each rank owns a
contiguous range of A rows and of B rows in its own HIP session, the updated shards travel between the processes
(gloo broadcasts of device tensors; RCCL refuses two ranks on one device), and the result must equal the
single-process run_poismf() bit for bit -- row results do not depend on how rows are cut into shards and launches, and
the column sums are recomputed from the replicated factor in a fixed order.  Needs an MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest

from poismf_amd import harness
from tests import helpers as H
from tests.test_gpu_parity import gpu_run

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("method,prec,early,shared_colsum", [("pg", "f32", False, True), ("cg", "f64", False, True), ("cg", "f32", False, True),
                                                             ("tncg", "f64", True, True), ("pg", "f32", False, False)])
def test_two_processes_reproduce_the_single_process_result(method, prec, early, shared_colsum, tmp_path):
    """shared_colsum: the first stage of the column sums cut between the ranks and its partials exchanged (poismf_amd/dist.py,
    _shared_colsum; forced here for factors of any size) -- the same bits as every rank summing the whole replica (the other case)."""
    out = str(tmp_path / "ranks.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               POISMF_SHARD_COLSUM_MIN_ROWS="1" if shared_colsum else str(10 ** 9))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tests", "dist2_worker.py"), out, method, prec] + (["early"] if early else [])
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=900)
    got = np.load(out)
    use_float = prec == "f32"
    csr, csc, A0, B0 = H.small_problem(3000, 2000, 120000, 50, use_float, seed=5, powerlaw=True, empty_rows=(3, 2999))
    l2, maxupd, _ = harness.auto_defaults(method, 50)
    kw = dict(maxupd=60) if method == "tncg" else {}
    A, B, _ = gpu_run(csr, csc, A0, B0, method, 3, 50, early_stop=early, reuse_prev=True, **kw)   # early: the summed counter
    assert np.array_equal(np.isfinite(got["A"]), np.isfinite(A))
    fa, fb = np.isfinite(A), np.isfinite(B)
    assert np.array_equal(got["A"][fa], A[fa]) and np.array_equal(got["B"][fb], B[fb])


def test_bench_gpus_2_launches_itself_and_prints_one_parseable_line():
    """`python3 bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset): the parent starts two fresh ranks before any GPU call,
    relays rank 0's compact line as its LAST stdout line and returns the ranks' exit code.  Two ranks on the one GPU of the test box
    (POISMF_BENCH_SHARE_GPUS=1, gloo: RCCL refuses two ranks on one device), the metric's matrix shrunk 10 x."""
    import json
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(POISMF_BENCH_SHARE_GPUS="1", POISMF_BENCH_BACKEND="gloo", POISMF_BENCH_SCALE="10", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    last = res.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and line["scaling"] == "strong"
    assert line["comm"]["ranks"] == 2 and line["comm"]["backend"] == "gloo" and line["rccl_ranks"] is None   # (gloo here; nccl reports RCCL's count)
    nnz = line["comm"]["nnz_per_rank"]
    assert len(nnz) == 2 and abs(sum(nnz) * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"]) / line["value"] - 1) < 1e-3
    assert line["comm"]["exchange_ms"] > 0
    assert line["roofline"]["frac"] > 0 and "cpu_baseline" not in line   # (the CPU leg is rank 0 at N = 1 only)
