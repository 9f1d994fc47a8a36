"""One-process-per-GPU driver for the alternation.

Rows of the factor being updated are independent given the full opposing factor, so each rank owns a
contiguous range of A rows (a CSR slice) and a contiguous range of B rows (a CSC slice); both factors are
replicated.  After each half-sweep the just-updated shard is all-gathered into every replica (RCCL over
xGMI on the GPU, gloo in the CPU tests); the k-length column sums are recomputed locally from the
replica, which is deterministic and needs no further collective.

The per-rank compute is a *backend* with three methods::

    half_sweep(which, step_size, cnst_div) -> n_unchanged   # updates its shard of B (which=0) / A (which=1) in place
    factor(which) -> torch.Tensor                            # the replicated factor this half updates, [dim x k]
    shard(which) -> (begin, end)

The product backend is HipBackend (the C-ABI session, no fallback).  Tests inject their own.
"""
import contextlib
import os

import numpy as np
import torch
import torch.distributed as dist

from . import api


def balanced_ranges(indptr, nparts):
    """Contiguous row ranges with (nearly) equal nonzero counts: cut points at the nnz quantiles of the
    prefix sums (balance nnz, not rows -- power-law item degrees)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    n = len(indptr) - 1
    total = int(indptr[-1])
    cuts = [0]
    for p in range(1, nparts):
        target = total * p // nparts
        c = int(np.searchsorted(indptr, target, side="left"))
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(nparts)]


def equal_ranges(n, nparts):
    """Contiguous ranges of (nearly) equal row counts (exactly equal when nparts divides n: enables the
    single-collective all_gather_into_tensor path)."""
    base, rem = divmod(n, nparts)
    out, s = [], 0
    for p in range(nparts):
        e = s + base + (1 if p < rem else 0)
        out.append((s, e))
        s = e
    return out


def exchange_shards(full, ranges, rank, group=None):
    """All-gather: on return every rank's `full` ([dim x k], replicated) holds every rank's row range.
    Equal ranges -> one in-place all_gather_into_tensor; otherwise one broadcast per owner."""
    world = dist.get_world_size(group)
    if world == 1 and os.environ.get("POISMF_BENCH_FORCE_DIST") != "1":
        return
    sizes = {e - b for b, e in ranges}
    per_owner = full.is_cuda and dist.get_backend(group) == "gloo"   # gloo has no all_gather_into_tensor on device memory
    if not per_owner and len(sizes) == 1 and ranges[0][0] == 0 and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1)):
        b, e = ranges[rank]
        # staged through a copy of the shard rather than gathered in place: the extra copy is the size of one
        # shard (C4 A half: 25 MB, ~10 us of HBM time) and avoids relying on in-place aliasing rules
        dist.all_gather_into_tensor(full, full[b:e].clone(), group=group)
        return
    for owner, (b, e) in enumerate(ranges):
        if e > b:
            dist.broadcast(full[b:e], src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)


class HipBackend:
    """The HIP session of this rank; factors are exposed as torch tensors aliasing the session's HBM."""

    def __init__(self, csr, csc, dimA, dimB, k, use_float, params_kw, shardA, shardB, device):
        torch.cuda.set_device(device)
        # A stream of its own, shared by the session's kernels and (through stream_context) by every collective on
        # the factors: the handle of torch's default stream is 0, which the C-ABI reads as "no stream given" and
        # answers with a private non-blocking stream -- unordered against the default stream the collectives would run on.
        self.stream = torch.cuda.Stream(device=device)
        self.sess = api.Session(csr, csc, dimA, dimB, k, use_float, device=device, stream=self.stream.cuda_stream,
                                shardA=shardA, shardB=shardB)
        self.params = self.sess.make_params(**params_kw)
        a, b = self.sess.device_arrays()
        self._A = torch.as_tensor(a, device=f"cuda:{device}")
        self._B = torch.as_tensor(b, device=f"cuda:{device}")
        self._shards = (tuple(shardB), tuple(shardA))

    def half_sweep(self, which, step_size, cnst_div, want_unchanged=False):
        return self.sess.half_sweep(which, self.params, step_size, cnst_div, want_unchanged)

    def stream_context(self):
        """Run torch operations on the factors (shard exchanges, reductions) in the session's stream order."""
        return torch.cuda.stream(self.stream)

    def factor(self, which):
        return self._A if which else self._B

    def shard(self, which):
        return self._shards[which]

    def close(self):
        self.sess.close()


class ShardedAlternation:
    """The reference's outer loop (ref: src/poismf.c:506-608) over a row-sharded backend."""

    def __init__(self, backend, rangesA, rangesB, method, l2_reg, step_size=1e-7, early_stop=False, dims=None,
                 group=None):
        self.be = backend
        self.ranges = (list(rangesB), list(rangesA))
        self.method = method
        self.l2_reg = float(l2_reg)
        self.step = float(step_size)
        self.early_stop = bool(early_stop) and method == "tncg"
        self.dims = dims  # (dimA, dimB), needed for the early-stop ratio
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.stopped = [False, False]  # [B, A]

    def _half(self, which, cnst_div):
        if self.method == "tncg" and self.stopped[which]:
            return
        n = self.be.half_sweep(which, self.step, cnst_div, self.early_stop)
        ctx = getattr(self.be, "stream_context", None)
        with (ctx() if ctx is not None else contextlib.nullcontext()):
            if dist.is_initialized():
                exchange_shards(self.be.factor(which), self.ranges[which], self.rank, self.group)
            if self.early_stop:  # ref: src/poismf.c:395-403, summed over shards
                t = torch.tensor([float(n)], dtype=torch.float64, device=self.be.factor(which).device)
                if dist.is_initialized():
                    dist.all_reduce(t, group=self.group)
                dim = self.dims[0] if which else self.dims[1]
                self.stopped[which] = (float(t.item()) / float(dim)) >= .95

    def sweep(self):
        """One full outer iteration; returns False once TNCG early stopping has ended both halves."""
        cnst_div = 1. / (1. + 2. * self.l2_reg * self.step)  # quirk Q6
        self._half(0, cnst_div)                              # B first (quirk Q5)
        if self.method == "pg":
            self.step *= 0.5
        self._half(1, cnst_div)
        return not (self.stopped[0] and self.stopped[1])
