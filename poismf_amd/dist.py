"""One-process-per-GPU driver for the alternation.

Rows of the factor being updated are independent given the full opposing factor, so each rank owns a
contiguous range of A rows (a CSR slice) and a contiguous range of B rows (a CSC slice); both factors are
replicated.  After each half-sweep the just-updated shard travels to every other replica (RCCL over xGMI on
the GPU, gloo in the CPU tests).  The k-length column sums of the fixed factor: every rank needs the same bits, and
a k-vector all-reduce over partial sums of row SHARDS would make them depend on the sharding; instead the sum is cut
into fixed blocks whose partial sums depend on the block number alone (include/poismf_hip.h,
poismf_hip_session_colsum_partial), each rank computes its 1/W of the blocks over the whole replica, the
[blocks x k] partials are all-gathered (100-400 KB) and every rank runs the fixed-order second stage: the unsharded
sum bit for bit, with the first stage's 1 / W per rank instead of all of it (round 4 recomputed the whole sum on
every rank: at C4 on 8 GPUs 50 us of a 650 us half).  Factors below SHARD_COLSUM_MIN_ROWS rows are summed locally.

The exchange is written for xGMI's full mesh, not for a ring: every rank sends its rows DIRECTLY to each of its
W - 1 peers and receives theirs straight into place in its replica -- one grouped launch of 2 (W - 1) point-to-point
operations (`batch_isend_irecv`: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd), all 7 links of a GPU busy at
once, no staging copy, any shard sizes (nnz-balanced ranges are unequal).  Equal contiguous shards take the
equivalent single in-place ncclAllGather.  A half-sweep may be cut into SEGMENTS (contiguous sub-ranges of the
rank's rows, each with its own launches): segment j's rows are exchanged on a second stream while segment j + 1
computes, so only the last segment's transfer is exposed.

The per-rank compute is a *backend* with these methods::

    half_sweep(which, step_size, cnst_div, want_unchanged=False, seg=None) -> n_unchanged
                                        # updates its shard of B (which=0) / A (which=1) in place; seg = j: only
                                        # segment j (the first segment also computes the column sums)
    factor(which) -> torch.Tensor       # the replicated factor this half updates, [dim x k]
    shard(which) -> (begin, end)
    segments(which) -> int              # optional (default 1)
    real(v) -> float                    # optional: v rounded to the backend's real type

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
    single in-place all-gather)."""
    base, rem = divmod(n, nparts)
    out, s = [], 0
    for p in range(nparts):
        e = s + base + (1 if p < rem else 0)
        out.append((s, e))
        s = e
    return out


def choose_ranges(counts, nparts, tolerance=0.03):
    """Row ranges for `nparts` ranks from per-row nonzero counts: equal row counts when that leaves the nonzeros
    balanced within `tolerance` (uniform data: the exchange is then one in-place all-gather), nnz-balanced cuts
    otherwise (power-law data)."""
    counts = np.asarray(counts, dtype=np.int64)
    indptr = np.concatenate([[0], np.cumsum(counts)])
    eq = equal_ranges(len(counts), nparts)
    loads = [int(indptr[e] - indptr[b]) for b, e in eq]
    if max(loads) <= (1.0 + tolerance) * (int(indptr[-1]) / nparts) + 1:
        return eq
    return balanced_ranges(indptr, nparts)


def segment_of(rng, j, nseg):
    """rows of segment j (of nseg) of the contiguous range rng = (begin, end): equal row counts"""
    b, e = rng
    n = e - b
    return b + n * j // nseg, b + n * (j + 1) // nseg


def exchange_shards(full, parts, rank, group=None):
    """`parts[r]` = (begin, end): the rows of the replicated `full` ([dim x k]) that rank r has just updated.  On return
    (in stream order for device tensors) every rank's `full` holds every rank's rows."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    backend = dist.get_backend(group)
    sizes = {e - b for b, e in parts}
    tiled = all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
    if backend == "nccl" and len(sizes) == 1 and tiled and parts[0][1] > parts[0][0]:
        # equal contiguous shards: ONE in-place all-gather (send buffer = this rank's slot of the receive buffer)
        b, e = parts[rank]
        dist.all_gather_into_tensor(full[parts[0][0]:parts[-1][1]], full[b:e], group=group)
        return
    if backend == "gloo" and full.is_cuda:
        # testing only (two processes on one GPU): gloo moves device tensors by broadcast, not by send / recv
        for owner, (b, e) in enumerate(parts):
            if e > b:
                dist.broadcast(full[b:e], src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
        return
    # any sizes: direct point-to-point, every pair at once, received in place
    ops = []
    mine = full[parts[rank][0]:parts[rank][1]]
    for peer, (b, e) in enumerate(parts):
        if peer == rank:
            continue
        gp = dist.get_global_rank(group, peer) if group is not None else peer
        if e > b:
            ops.append(dist.P2POp(dist.irecv, full[b:e], gp, group))
        if mine.shape[0] > 0:
            ops.append(dist.P2POp(dist.isend, mine, gp, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


SHARD_COLSUM_MIN_ROWS = int(os.environ.get("POISMF_SHARD_COLSUM_MIN_ROWS", "262144"))   # (a 1e5-row factor sums in 5 us: not worth a collective)


class HipBackend:
    """The HIP session of this rank; factors are exposed as torch tensors aliasing the session's HBM."""

    def __init__(self, csr, csc, dimA, dimB, k, use_float, params_kw, shardA, shardB, device, coo=None, segments=(1, 1)):
        torch.cuda.set_device(device)
        # A stream of its own, shared by the session's kernels and (through stream_context) by the collectives on
        # the factors: the C-ABI reads a null stream handle (torch's default stream) as "no stream given" and answers
        # with a private non-blocking stream -- unordered against the default stream the collectives would run on.
        self.stream = torch.cuda.Stream(device=device)
        self.comm_stream = torch.cuda.Stream(device=device)
        if coo is not None:
            self.sess = api.Session.from_coo(coo, k, use_float, device=device, stream=self.stream.cuda_stream, shardA=shardA,
                                             shardB=shardB)
        else:
            self.sess = api.Session(csr, csc, dimA, dimB, k, use_float, device=device, stream=self.stream.cuda_stream,
                                    shardA=shardA, shardB=shardB)
        self.params = self.sess.make_params(**params_kw)
        a, b = self.sess.device_arrays()
        self._A = torch.as_tensor(a, device=f"cuda:{device}")
        self._B = torch.as_tensor(b, device=f"cuda:{device}")
        self._shards = (tuple(shardB), tuple(shardA))
        self._nseg = [1, 1]
        for which in (0, 1):
            if segments[which] > 1:
                self._nseg[which] = self.sess.set_segments(which, segments[which])

    def half_sweep(self, which, step_size, cnst_div, want_unchanged=False, seg=None):
        return self.sess.half_sweep(which, self.params, step_size, cnst_div, want_unchanged, seg)

    def stream_context(self):
        """Run torch operations on the factors (shard exchanges, reductions) in the session's stream order."""
        return torch.cuda.stream(self.stream)

    def factor(self, which):
        return self._A if which else self._B

    def shard(self, which):
        return self._shards[which]

    def segments(self, which):
        return self._nseg[which]

    def real(self, v):
        return self.sess.real(v)

    def factors_dirty(self, which):
        self.sess.factors_dirty(which)

    # -- the first stage of the column sums, shared between the ranks (module docstring) --
    def colsum_rows(self, which):
        """rows of the FIXED factor of half `which` (A for the B half, B for the A half)"""
        return self.sess.dimB if which else self.sess.dimA

    def colsum_blocks(self, which):
        return self.sess.colsum_blocks(which)

    def colsum_partial(self, which, b_lo, b_hi):
        self.sess.colsum_partial(which, b_lo, b_hi)

    def partials(self, which):
        return torch.as_tensor(self.sess.partials_array(which), device=self._A.device)

    def partials_ready(self):
        self.sess.partials_ready()

    def close(self):
        self.sess.close()


class ShardedAlternation:
    """The reference's outer loop (ref: src/poismf.c:506-608) over a row-sharded backend."""

    def __init__(self, backend, rangesA, rangesB, method, l2_reg, step_size=1e-7, early_stop=False, dims=None,
                 group=None):
        self.be = backend
        self.ranges = (list(rangesB), list(rangesA))
        self.method = method
        self._real = getattr(backend, "real", float)
        self.l2_reg = self._real(l2_reg)
        self.step = self._real(step_size)
        self.early_stop = bool(early_stop) and method == "tncg"
        self.dims = dims  # (dimA, dimB), needed for the early-stop ratio
        self.group = group
        self.multi = dist.is_initialized() and dist.get_world_size(group) > 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.stopped = [False, False]  # [B, A]
        self._timed = None  # measurement only (bench.py): [(what, start event, end event)] of every exchange while switched on

    def time_exchanges(self, on=True):
        """Bracket every exchange with device events on the stream it is issued on (a profiled pass of bench.py; off in the product path)."""
        self._timed = [] if on else None

    def exchange_ms(self):
        """{"rows": ms, "colsum_partials": ms, "calls": n} of the exchanges since time_exchanges(True): device time between each exchange's
        two events (on its own stream: what the transfer occupies, whether or not compute on the other stream hides it)."""
        out = {"rows": 0.0, "colsum_partials": 0.0, "calls": 0}
        for what, e0, e1 in self._timed or []:
            e1.synchronize()
            out[what] += e0.elapsed_time(e1)
            out["calls"] += 1
        return out

    @contextlib.contextmanager
    def _bracket(self, what, tensor):
        if self._timed is None or not tensor.is_cuda:
            yield
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        yield
        e1.record()
        self._timed.append((what, e0, e1))

    def _exchange(self, which, j, nseg):
        parts = [segment_of(r, j, nseg) for r in self.ranges[which]]
        full = self.be.factor(which)
        with self._bracket("rows", full):
            exchange_shards(full, parts, self.rank, self.group)

    def _shared_colsum(self, which, ctx):
        """Each rank computes its share of the blocks of the fixed factor's column sums, the partials are all-gathered (in the
        session's stream order: the half-sweep that follows reads them), and the backend is told they are complete."""
        be = self.be
        if not (self.multi and hasattr(be, "colsum_partial") and be.colsum_rows(which) >= SHARD_COLSUM_MIN_ROWS):
            return
        world = dist.get_world_size(self.group)
        nb = be.colsum_blocks(which)
        parts = equal_ranges(nb, world)
        be.colsum_partial(which, *parts[self.rank])
        with (ctx() if ctx is not None else contextlib.nullcontext()):
            part = be.partials(which)
            with self._bracket("colsum_partials", part):
                exchange_shards(part, parts, self.rank, self.group)
        be.partials_ready()

    def _half(self, which, cnst_div):
        if self.method == "tncg" and self.stopped[which]:
            return
        nseg = getattr(self.be, "segments", lambda w: 1)(which)
        ctx = getattr(self.be, "stream_context", None)
        comm = getattr(self.be, "comm_stream", None)
        n = 0
        self._shared_colsum(which, ctx)
        if nseg == 1:
            n = self.be.half_sweep(which, self.step, cnst_div, self.early_stop)
            with (ctx() if ctx is not None else contextlib.nullcontext()):
                if self.multi:
                    self._exchange(which, 0, 1)
        elif comm is None or not self.multi:
            # segments without a second stream (CPU backends of the tests, single rank): same order of work, no overlap
            for j in range(nseg):
                n = self.be.half_sweep(which, self.step, cnst_div, self.early_stop and j == nseg - 1, seg=j)
                with (ctx() if ctx is not None else contextlib.nullcontext()):
                    if self.multi:
                        self._exchange(which, j, nseg)
        else:
            # segment j's rows travel on the communication stream while segment j + 1 computes
            for j in range(nseg):
                n = self.be.half_sweep(which, self.step, cnst_div, self.early_stop and j == nseg - 1, seg=j)
                done = torch.cuda.Event()
                done.record(self.be.stream)
                comm.wait_event(done)
                with torch.cuda.stream(comm):
                    self._exchange(which, j, nseg)
            landed = torch.cuda.Event()
            landed.record(comm)
            self.be.stream.wait_event(landed)   # the next half reads the whole factor
        if self.multi and hasattr(self.be, "factors_dirty"):
            self.be.factors_dirty(which)
        if self.early_stop:  # ref: src/poismf.c:395-403, summed over shards
            with (ctx() if ctx is not None else contextlib.nullcontext()):
                t = torch.tensor([float(n)], dtype=torch.float64, device=self.be.factor(which).device)
                if self.multi:
                    dist.all_reduce(t, group=self.group)
                dim = self.dims[0] if which else self.dims[1]
                self.stopped[which] = (float(t.item()) / float(dim)) >= .95

    def sweep(self):
        """One full outer iteration; returns False once TNCG early stopping has ended both halves."""
        # quirk Q6; evaluated as run_poismf does: real_t operands, double expression, real_t result (ref: src/poismf.c:511)
        cnst_div = self._real(1. / (1. + 2. * self.l2_reg * self.step))
        self._half(0, cnst_div)                              # B first (quirk Q5)
        if self.method == "pg":
            self.step = self._real(self.step * 0.5)
        self._half(1, cnst_div)
        return not (self.stopped[0] and self.stopped[1])
