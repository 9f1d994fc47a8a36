"""The N > 1 path on CPU: world_size-2 gloo processes run poismf_amd.dist.ShardedAlternation with a TEST
backend (the oracle restricted to a row shard) and must reproduce the unsharded oracle bit for bit.  This
covers the partitioning, the shard exchange (equal and unequal ranges), the step schedule and the TNCG
early-stop reduction -- everything of the multi-GPU driver except the HIP kernels themselves."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bindings
from poismf_amd import dist as pdist
from poismf_amd import harness
from tests import helpers as H


class OracleShardBackend:
    """Per-rank compute for the CPU test: the oracle's half-sweep drivers on this rank's rows only."""

    def __init__(self, csr, csc, A0, B0, method, l2, maxupd, shardA, shardB, is_float, early_stop=False, nseg=(1, 1)):
        self.orc = bindings.Oracle(is_float)
        self.csr, self.csc = csr, csc
        self.A, self.B = torch.from_numpy(A0.copy()), torch.from_numpy(B0.copy())
        self.method, self.l2, self.maxupd = method, l2, maxupd
        self.shards = (shardB, shardA)
        self.early_stop = early_stop
        self.nseg = nseg
        self._unchanged = 0

    def segments(self, which):
        return self.nseg[which]

    def factor(self, which):
        return self.A if which else self.B

    def shard(self, which):
        return self.shards[which]

    def half_sweep(self, which, step, cnst_div, want_unchanged=False, seg=None):
        """seg = j: only segment j of the shard (cut as the C session cuts it); the unchanged-row count accumulates over
        the segments of a half and is reported by the last one"""
        M = (self.A if which else self.B).numpy()
        F = (self.B if which else self.A).numpy()
        data, indices, indptr = self.csr if which else self.csc
        b, e = self.shards[which]
        if seg is not None:
            b, e = pdist.segment_of((b, e), seg, self.nseg[which])
            if seg == 0:
                self._unchanged = 0
        else:
            self._unchanged = 0
        ptr = (indptr[b:e + 1] - indptr[b]).astype(np.uint64)
        sl = slice(int(indptr[b]), int(indptr[e]))
        Ms = np.ascontiguousarray(M[b:e])
        bs = self.orc.sum_by_cols(F)
        n = 0
        if self.method == "pg":
            cs = bs * np.asarray(-step, bs.dtype)
            if which:
                cs = cs * np.asarray(-step, bs.dtype)
            self.orc.pg_iteration(Ms, F, data[sl], ptr, indices[sl], cnst_div, cs, None, step, 1.0, self.maxupd)
        elif self.method == "cg":
            self.orc.cg_iteration(Ms, F, data[sl], ptr, indices[sl], True, bs, self.l2, 1.0, self.maxupd)
        else:
            Mp = Ms.copy()
            self.orc.tncg_iteration(Ms, F, False, data[sl], ptr, indices[sl], bs, self.l2, 1.0, self.maxupd, False)
            nz = np.diff(ptr.astype(np.int64)) > 0
            n = int(np.sum((((Mp - Ms) ** 2).sum(1) <= 1e-4) & nz))
        M[b:e] = Ms
        self._unchanged += n
        return self._unchanged


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(shape):
    if shape == "c5":   # config-C5-shaped: power-law item degrees, short user rows: the nnz-balanced ranges of eight ranks are very unequal
        return H.small_problem(400, 300, 9000, 6, False, seed=13, powerlaw=True, empty_rows=(4, 250)), (400, 300)
    return H.small_problem(90, 70, 1500, 6, False, seed=11, powerlaw=True, empty_rows=(4,)), (90, 70)


def _worker(rank, world, port, method, is_float, balanced, numiter, outdir, nseg=(1, 1), shape="small"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    (csr, csc, A0, B0), (dimA, dimB) = _problem(shape)
    l2, maxupd, _ = harness.auto_defaults(method, 6)
    if method == "tncg":
        maxupd = 40
    if balanced:
        rA, rB = pdist.balanced_ranges(csr[2], world), pdist.balanced_ranges(csc[2], world)
    else:
        rA, rB = pdist.equal_ranges(dimA, world), pdist.equal_ranges(dimB, world)
    be = OracleShardBackend(csr, csc, A0, B0, method, l2, maxupd, rA[rank], rB[rank], is_float, nseg=nseg)
    alt = pdist.ShardedAlternation(be, rA, rB, method, l2, 1e-7, early_stop=(method == "tncg"), dims=(dimA, dimB))
    for _ in range(numiter):
        if not alt.sweep():
            break
    np.savez(os.path.join(outdir, f"r{rank}.npz"), A=be.A.numpy(), B=be.B.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("method", ["pg", "cg", "tncg"])
@pytest.mark.parametrize("balanced,nseg", [(False, (1, 1)), (True, (1, 1)), (True, (2, 3))])
def test_world2_gloo_matches_unsharded_oracle(tmp_path, method, balanced, nseg):
    """equal ranges, nnz-balanced (unequal) ranges, and halves cut into segments that are exchanged one by one (the
    point-to-point exchange of poismf_amd/dist.py: every rank sends its rows to its peer and receives in place)"""
    is_float = False
    world, numiter = 2, 2
    mp.spawn(_worker, args=(world, _free_port(), method, is_float, balanced, numiter, str(tmp_path), nseg), nprocs=world, join=True)
    csr, csc, A0, B0 = H.small_problem(90, 70, 1500, 6, is_float, seed=11, powerlaw=True, empty_rows=(4,))
    l2, maxupd, _ = harness.auto_defaults(method, 6)
    if method == "tncg":
        maxupd = 40
    A, B = A0.copy(), B0.copy()
    bindings.Oracle(is_float).run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], l2, 0.0, 1.0, 1e-7, method,
                                         True, numiter, maxupd, method == "tncg", False)
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["A"], A) and np.array_equal(z["B"], B)


@pytest.mark.parametrize("method", ["pg", "tncg"])
def test_world8_gloo_on_power_law_ranges_matches_unsharded_oracle(tmp_path, method):
    """The node the north-star names has eight GPUs: EIGHT gloo ranks on a config-C5-shaped matrix (power-law item degrees: the
    nnz-balanced B ranges run from a handful of rows to hundreds), the A half exchanged in two segments -- the 56 point-to-point
    transfers of a half, unequal (and possibly empty) shards, the summed early-stop counter.  Bit for bit the unsharded oracle."""
    world, numiter = 8, 2
    mp.spawn(_worker, args=(world, _free_port(), method, False, True, numiter, str(tmp_path), (1, 2), "c5"), nprocs=world, join=True)
    (csr, csc, A0, B0), _ = _problem("c5")
    rB = pdist.balanced_ranges(csc[2], world)
    sizes = [e - b for b, e in rB]
    assert max(sizes) >= 8 * max(1, min(sizes))          # (the ranges ARE very unequal: what this test is for)
    l2, maxupd, _ = harness.auto_defaults(method, 6)
    if method == "tncg":
        maxupd = 40
    A, B = A0.copy(), B0.copy()
    bindings.Oracle(False).run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], l2, 0.0, 1.0, 1e-7, method,
                                      True, numiter, maxupd, method == "tncg", False)
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["A"], A) and np.array_equal(z["B"], B)


def test_balanced_ranges_cover_and_balance():
    rng = np.random.default_rng(0)
    nnz_per_row = (rng.pareto(1.2, 5000) * 10).astype(np.int64)
    indptr = np.concatenate([[0], np.cumsum(nnz_per_row)])
    for parts in (2, 3, 8):
        r = pdist.balanced_ranges(indptr, parts)
        assert r[0][0] == 0 and r[-1][1] == 5000 and all(r[i][1] == r[i + 1][0] for i in range(parts - 1))
        loads = [indptr[e] - indptr[b] for b, e in r]
        assert max(loads) <= indptr[-1] / parts + nnz_per_row.max()
    assert pdist.equal_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
