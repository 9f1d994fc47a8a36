#!/usr/bin/env python3
"""bench.py -- nonzeros/sec per full A+B sweep of the factor-update hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--method pg|cg|tncg] [--maxupd M] [--fp64] [--no-cpu] [--no-extra]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full outer iteration (B half + A half, column sums and -- multi-GPU -- the shard exchanges
included) on synthetic data that is already resident in HBM when the timed region starts.

Workload (every N): the matrix BASELINE.json quotes the metric on -- uniform 1 000 000 x 100 000, 1e8 triplets
(seed 1; ~9.995e7 nonzeros after duplicate summing), k = 50 (configs C3 / C4).  The headline line is method = pg,
fp32, the reference's Python defaults (l2 1e9, step 1e-7, maxupd 10); at N = 1 `extra` carries the other points of
the metric on the SAME matrix, each with its own roofline block: pg with maxupd = 1 (the bandwidth point, R's
default), pg with hyper-parameters that keep the factors finite (same work per sweep), cg and tncg fp32 (the solvers config
C4's matrix is not quoted on, for completeness), cg fp64 (config C3) -- and `tncg_f64_c5`: config C5's own matrix (Last.FM-shaped
358 858 x 160 112, power-law item degrees, k = 100, tncg fp64, l2 1e3, maxupd 1500), steady-state sweeps.  Every block carries the
HBM roofline of SURVEY.md 8(d) AND a vector-ALU roofline (`roofline.valu`: the flops the reference's arithmetic needs for the
decisions the solvers took, against the fp32 / fp64 vector peak) -- the multi-pass solvers are not HBM-bound.
N > 1 is STRONG scaling of that one matrix: A rows and B rows are cut into per-rank ranges with balanced nonzero
counts, every rank builds its CSR / CSC shards on its GPU from the triplets, both factors are replicated, and
after each half the updated rows go to every peer over RCCL (poismf_amd/dist.py).

Rank 0 prints ONE JSON line (see the driver contract); `roofline` prices the row-update kernels against HBM
peak with SURVEY.md 8(d)'s algorithmic bytes, `cpu_baseline` is the compiled reference (oracle/_ref, kind
"reference") timed on this box's host cores on the same matrix (steady state: the difference of a 2-iteration
and a 1-iteration run).  `roofline.by_config` / `cpu_baseline.by_config` put the three lines of the metric side by side:
PG with the reference's defaults (the headline), PG with hyper-parameters that keep the factors alive, CG fp64.
"""
import argparse
import json
import os
import sys
import time

_AFFINITY = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None   # (before libgomp binds this thread, below)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")  # the CPU baseline threads over rows with OpenMP, as the reference does
# the CPU baseline runs with its OpenMP threads bound (read by libgomp when it is loaded: before numpy / torch)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from poismf_amd import api, build, harness, synth  # noqa: E402
from poismf_amd import dist as pdist  # noqa: E402

# With OMP_PROC_BIND set, libgomp pins the thread that loads it -- this one -- to its first place, and every thread created from it
# later inherits that one-core mask: the eight host threads that fill run_poismf's pinned upload chunks then share a core (measured:
# the drop-in call's first iteration 80 -> 97 ms).  The main thread gets its mask back; the OpenMP workers of the CPU legs are still
# bound to their places when their team is created.
if _AFFINITY is not None:
    try:
        os.sched_setaffinity(0, _AFFINITY)
    except OSError:
        pass

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOADS = {
    # name: (dimA, dimB, triplets)
    "C4": (10 ** 6, 10 ** 5, 10 ** 8),   # = C3's matrix; the one BASELINE.json's metric is quoted on
    "C2": (10 ** 5, 10 ** 5, 10 ** 7),   # development only
}
K = 50
C5_K, C5_MAXUPD, C5_L2 = 100, 1500, 1e3   # BASELINE config 5: tncg fp64, k = 100, maxupd = 15 k, l2 = the tncg default


def algorithmic_bytes_half(nnz, dimM, k, s):
    """one gather per stored nonzero; device widths s_idx = 4, s_ptr = 8."""
    return nnz * (4 + s + k * s) + 2 * dimM * k * s + (dimM + 1) * 8


def plan_ranges(trip, world):
    """Per-rank row ranges of A and of B, identical on every rank (computed from the seeded triplets)."""
    dimA, dimB = trip.shape
    if world == 1:
        return [(0, dimA)], [(0, dimB)]
    return (pdist.choose_ranges(np.bincount(trip.row, minlength=dimA), world),
            pdist.choose_ranges(np.bincount(trip.col, minlength=dimB), world))


def timed_sweeps(alt, steps, device):
    multi = dist.is_initialized()
    if multi:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        alt.sweep()
    torch.cuda.synchronize(device)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{device}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


class Job:
    """One session (one precision) on this rank's shards, reusable for several (method, hyper-parameter) runs."""

    def __init__(self, trip, rangesA, rangesB, rank, device, use_float, segmentsA, k=K):
        self.trip, self.rangesA, self.rangesB, self.rank, self.device, self.use_float = trip, rangesA, rangesB, rank, device, use_float
        self.dimA, self.dimB = trip.shape
        self.k = k
        t0 = time.perf_counter()
        self.be = pdist.HipBackend(None, None, self.dimA, self.dimB, k, use_float,
                                   dict(method="pg", l2_reg=1.0), rangesA[rank], rangesB[rank], device, coo=trip,
                                   segments=(1, segmentsA))
        torch.cuda.synchronize(device)
        self.setup_s = time.perf_counter() - t0
        self.A0, self.B0 = harness.initialize_matrices(self.dimA, self.dimB, k, use_float, 1)
        self.nnz_local = (self.be.sess.nnz(0), self.be.sess.nnz(1))

    def run(self, method, maxupd, steps, warmup, l2=None, step0=1e-7, profile_steps=None):
        """warmup + `steps` timed sweeps with profiling OFF (the product configuration), then a separate profiled pass
        (HIP events around the row-kernel launches of each half, per-row pass counters) of `profile_steps` sweeps from
        the same starting point for the kernel time and the pass-weighted traffic."""
        l2d, mu, _ = harness.auto_defaults(method, self.k)
        l2 = l2d if l2 is None else l2
        maxupd = mu if maxupd is None else maxupd
        sess = self.be.sess
        self.be.params = sess.make_params(method=method, l2_reg=l2, maxupd=maxupd, limit_step=True, early_stop=False, reuse_prev=True)
        dims = (self.dimA, self.dimB)

        def fresh():
            sess.set_factors(self.A0, self.B0)
            return pdist.ShardedAlternation(self.be, self.rangesA, self.rangesB, method, l2, step0, dims=dims)

        alt = fresh()
        for _ in range(warmup):
            alt.sweep()
        dt = timed_sweeps(alt, steps, self.device)
        A, B = sess.get_factors()
        finite = bool(np.isfinite(A).all() and np.isfinite(B).all())
        # "finite" only says no NaN / inf; a factor the solver has driven to all zeros is finite too -- report how much is alive
        alive = {"A_nonzero_frac": float(np.count_nonzero(A)) / A.size, "B_nonzero_frac": float(np.count_nonzero(B)) / B.size}
        del A, B
        psteps = profile_steps or min(steps, 5)
        alt = fresh()
        for _ in range(warmup):
            alt.sweep()
        sess.profile(True)
        alt.time_exchanges(True)
        for _ in range(psteps):
            alt.sweep()
        k_ms = [sess.kernel_time(w) for w in (0, 1)]
        ex = alt.exchange_ms()
        alt.time_exchanges(False)
        exchange = {"rows_ms_per_sweep": ex["rows"] / psteps, "colsum_partials_ms_per_sweep": ex["colsum_partials"] / psteps,
                    "exchanges_per_sweep": ex["calls"] / psteps}
        ev_stats = [sess.eval_stats(w) for w in (0, 1)]
        dec_stats = [sess.decision_stats(w) for w in (0, 1)]   # (of the last profiled sweep)
        plan = [sess.plan(w) for w in (0, 1)]
        lprof = [sess.launch_profile(w) for w in (0, 1)]
        sess.profile(False)
        return dict(method=method, maxupd=maxupd, l2=l2, step0=step0, steps=steps, seconds=dt, finite=finite, alive=alive, psteps=psteps,
                    kernel_ms=k_ms, ev_stats=ev_stats, dec_stats=dec_stats, plan=plan, lprof=lprof, exchange=exchange)

    def close(self):
        self.be.close()


def roofline_block(job, res, traffic_key=None):
    """HBM roofline of one run, this rank's launches.  `achieved` / `frac`: SURVEY.md 8(d)'s algorithmic bytes of the rank's two
    shards per sweep / the UNPROFILED time of a sweep (the timed region: row kernels + column sums, profiling off).  The
    row-kernel durations (`kernel_ms_*`, `frac_row_kernels`, `launches`) come from a separate profiled pass: HIP events on the
    stream each launch is issued on, around every row-bin launch and around each half."""
    s = 4 if job.use_float else 8
    K = job.k
    rA, rB = job.rangesA[job.rank], job.rangesB[job.rank]
    b_half = [algorithmic_bytes_half(job.nnz_local[0], rB[1] - rB[0], K, s), algorithmic_bytes_half(job.nnz_local[1], rA[1] - rA[0], K, s)]
    k_ms_half = [res["kernel_ms"][w][0] / res["psteps"] for w in (0, 1)]
    k_ms = sum(k_ms_half)
    sweep_ms = res["seconds"] / res["steps"] * 1e3
    achieved = sum(b_half) / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    achieved_k = sum(b_half) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    rows = (rA[1] - rA[0]) + (rB[1] - rB[0])
    passes = res["ev_stats"][0][0] + res["ev_stats"][1][0]
    nnzp = res["ev_stats"][0][1] + res["ev_stats"][1][1]
    gb = nnzp * K * s / res["psteps"] / 1e9
    traffic, traffic_source = traffic_from_profiles(traffic_key)
    valu = valu_block(job, res, sweep_ms)
    # per launch: algorithmic bytes of the rows it covers / its average duration
    launches = []
    for w in (0, 1):
        for L in res["lprof"][w]:
            by = L["nnz"] * (4 + s + K * s) + 2 * L["rows"] * K * s + L["rows"] * 8
            ms = L["ms"] / max(L["calls"], 1)
            gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            launches.append({"kernel": L["kernel"], "half": "A" if w else "B", "rows": L["rows"], "nnz": L["nnz"], "avg_ms": ms,
                             "calls": L["calls"], "algorithmic_bytes": int(by), "GBps": gbs, "frac": gbs / HBM_PEAK_GBS})
    dom = max(launches, key=lambda L: L["avg_ms"]) if launches else None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": traffic_source, "valu": valu,
            "sweep_ms_unprofiled": sweep_ms,
            "dominant_kernel": dom,
            "frac_row_kernels": achieved_k / HBM_PEAK_GBS,
            "kernel_ms_per_sweep": k_ms, "kernel_ms_B_half": k_ms_half[0], "kernel_ms_A_half": k_ms_half[1],
            "launches": launches,
            "algorithmic_bytes_per_sweep": int(sum(b_half)), "half_sweeps_profiled": int(res["kernel_ms"][0][1] + res["kernel_ms"][1][1]),
            "pass_weighted": {"tile_passes_per_row": passes / res["psteps"] / max(rows, 1), "on_chip_GB_per_sweep": gb,
                              "on_chip_GBps": gb / (k_ms * 1e-3) if k_ms > 0 else 0.0},
            "note": "achieved / frac = algorithmic bytes of one sweep / the unprofiled wall time of one sweep (the timed region of this "
                    "block); frac_row_kernels, kernel_ms_* and launches[] come from a separate profiled pass (HIP events on the stream "
                    "each launch is issued on; launches are serial, so their sum equals sum(Calls x AverageNs) of the half_sweep_* rows "
                    "of the rocprofv3 kernel stats under profiles/); dominant_kernel = the launch with the longest average duration; "
                    "bytes = nnz*(4+s+k*s) + 2*dimM*k*s + (dimM+1)*8 per half; traffic = fabric-side bytes per sweep from separate rocprofv3 "
                    "--pmc passes of the same command (traffic_source says which committed file, and that its kernels are this tree's; "
                    "null when the tree has changed since); valu = the second roofline: SURVEY.md 8(d)'s flop count against the vector peak"}


VALU_PEAK_TFLOPS = {True: 157.3, False: 78.6}   # MI355X vector fp32 / fp64 (MI355X_MICROARCH.md); no MFMA on this path
LOG_FLOPS = 25                                 # a log counted as ~25 flop-equivalents


def valu_block(job, res, sweep_ms):
    """The vector-ALU roofline of one run ("Flops"): the flops the REFERENCE's arithmetic needs for the decisions the
    device's solvers took in the last profiled sweep -- a gradient (4k+1) per nonzero (ref src/poismf.c:126-133, :210-240), a
    function value (2k+L) (ref :194-208), TNC's fused evaluation (4k+1+L) (ref :242-273) -- divided by the unprofiled sweep time
    and by the vector peak of the precision.  The device does less than this where it prunes or caches line-search trials; the
    numerator is the algorithm's work, as `achieved` is the algorithm's bytes."""
    k = job.k
    method = res["method"]
    if method == "pg":
        # every pass of every row is one gradient evaluation (ev_stats: sum over rows of passes x nonzeros, profiled sweeps)
        nnzp = (res["ev_stats"][0][1] + res["ev_stats"][1][1]) / res["psteps"]
        flops = nnzp * (4 * k + 1)
        model = "sum_rows nnz x passes x (4k+1)"
    elif method == "cg":
        d = res["dec_stats"]
        g = d[0]["nnz_iterations"] + d[1]["nnz_iterations"]                       # one gradient per iteration
        f = g + d[0]["nnz_evaluations"] + d[1]["nnz_evaluations"]                 # f0 + failed trials (nfeval) + one accepted trial per iteration
        flops = g * (4 * k + 1) + f * (2 * k + LOG_FLOPS)
        model = "sum_rows nnz x (niter x (4k+1) + (nfeval + niter) x (2k+L)), L=25; niter / nfeval as minimize_nonneg_cg counts them"
    else:
        d = res["dec_stats"]
        flops = (d[0]["nnz_evaluations"] + d[1]["nnz_evaluations"]) * (4 * k + 1 + LOG_FLOPS)
        model = "sum_rows nnz x nfeval x (4k+1+L), L=25; nfeval as tnc counts fun_and_grad calls"
    peak = VALU_PEAK_TFLOPS[bool(job.use_float)]
    ach = flops / (sweep_ms * 1e-3) / 1e12 if sweep_ms > 0 else 0.0
    return {"bound": "valu", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "flops_per_sweep": flops, "model": model,
            "note": "reference-equivalent flops of the last profiled sweep / unprofiled sweep time / vector peak (fp32 157.3, fp64 78.6 TFLOP/s)"}


def traffic_from_profiles(key):
    """(bytes per sweep, where it came from) -- the fabric-side traffic of the row kernels measured by separate rocprofv3 --pmc passes
    (scripts/profile_round.sh) and committed under profiles/.  Only quoted when the file was recorded from THIS tree's kernels
    (its source_hash equals poismf_amd.build's hash of the sources + flags); otherwise null, and the reason."""
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not key or not os.path.exists(tf):
        return None, None
    try:
        d = json.load(open(tf))
    except Exception:
        return None, "profiles/hbm_traffic.json unreadable"
    if key not in d:
        return None, f"profiles/hbm_traffic.json has no entry {key}"
    have, want = d.get("source_hash"), build._source_hash()
    if have != want:
        return None, (f"profiles/hbm_traffic.json ({d.get('round', '?')}) was recorded from other kernel sources "
                      f"(source_hash {str(have)[:12]} != this tree's {want[:12]}): not quoted")
    return d[key], f"profiles/hbm_traffic.json@{d.get('round', '?')} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, source_hash {want[:12]} = this tree)"


def abi_timing(csr, csc, dimA, dimB, method, use_float, maxupd):
    """Wall time of the drop-in run_poismf() itself -- host arrays in, host arrays out, set-up (upload, index narrowing and
    row sort on the device) included: the PCIe-inclusive cost a caller of the reference's ABI sees.  Never `value`."""
    l2, mu, _ = harness.auto_defaults(method, K)
    maxupd = mu if maxupd is None else maxupd
    A0, B0 = harness.initialize_matrices(dimA, dimB, K, use_float, 1)
    def calls(seq):
        t = {1: [], 6: []}
        for numiter in seq:
            A, B = A0.copy(), B0.copy()
            t0 = time.perf_counter()
            with np.errstate(all="ignore"):
                api._run_poismf(csr[0], csr[1], csr[2], csc[0], csc[1], csc[2], A, B, method, True, l2, 0., 1., 1e-7, numiter, maxupd,
                                False, True, True, 1)
            t[numiter].append((time.perf_counter() - t0) * 1e3)
        return t

    # the library's default: nothing survives a call (every device array is freed before run_poismf returns, as the reference does)
    limit_before = api.set_device_cache_mb(0, use_float)   # ({flavour: MB}: what the user had asked for, e.g. POISMF_HIP_DEVICE_CACHE_MB; put back below)
    t = calls((1, 1, 6, 1, 6, 1, 6))   # (the first call also pays the pinned staging chunks: dropped)
    t1, t6 = min(t[1][1:]), min(t[6])
    # opted in to keeping released device arrays between calls (POISMF_HIP_DEVICE_CACHE_MB / poismf_hip_set_device_cache_mb)
    api.set_device_cache_mb(16384, use_float)
    tc = calls((1, 1, 1, 1))
    t1_cache = min(tc[1][1:])
    api.set_device_cache_mb(next(iter(limit_before.values())), use_float)
    # where a call's time goes: the same steps through the session entry points, each timed (min of 3): what run_poismf does inside
    split = {"session_create_upload_X_and_sort": [], "factors_up": [], "one_iteration": [], "factors_down": [], "destroy": []}
    outA, outB = A0.copy(), B0.copy()   # (touched pages, as run_poismf's own in / out arrays are)
    for _ in range(3):
        t0 = time.perf_counter()
        sess = api.Session(csr, csc, dimA, dimB, K, use_float)
        ta = time.perf_counter()
        sess.set_factors(A0, B0)
        tb = time.perf_counter()
        prm = sess.make_params(method=method, l2_reg=l2, maxupd=maxupd, limit_step=True, early_stop=False, reuse_prev=True)
        with np.errstate(all="ignore"):
            sess.sweep(prm, 1e-7)
        sess.kernel_time(0)   # (synchronises the session stream)
        tc = time.perf_counter()
        sess.get_factors(out=(outA, outB))
        td = time.perf_counter()
        sess.close()
        te = time.perf_counter()
        for name, dt in zip(split, (ta - t0, tb - ta, tc - tb, td - tc, te - td)):
            split[name].append(dt * 1e3)
    return {"abi_ms_first_iter": t1, "abi_ms_first_iter_cache_on": t1_cache, "abi_ms_per_extra_iter": (t6 - t1) / 5.0, "abi_ms_six_iters": t6,
            "samples_ms": {"numiter1": t[1][1:], "numiter6": t[6]}, "first_call_ms": t[1][0],
            "split_ms": {k_: min(v) for k_, v in split.items()},
            "note": f"run_poismf(method={method}, maxupd={maxupd}, {'fp32' if use_float else 'fp64'}) on the workload matrix through "
                    "ctypes: min of 3 calls each with numiter 1 and 6; per_extra_iter = (min t6 - min t1) / 5.  The process's first run_poismf "
                    "call (first_call_ms; the device is up by then, the blocks above ran first) also pays the pinned staging chunks.  abi_ms_first_iter "
                    "is the library's default (every device array freed before the call returns); abi_ms_first_iter_cache_on is the same call "
                    "after poismf_hip_set_device_cache_mb(16384): released arrays are kept for the next call (devmem.hpp)"}


_CPU_CACHE = {}
_CPU_THREADS = {}   # the OpenMP thread count the CPU legs use, chosen once per process (cpu_baseline)


def _cpu_inputs(trip, use_float, k=K):
    """host CSR / CSC (size_t indices) for the reference's ABI + the starting factors, once per (matrix, precision)"""
    key = (id(trip), use_float)
    if key not in _CPU_CACHE:
        for old in [q for q in _CPU_CACHE if isinstance(q, tuple)]:
            del _CPU_CACHE[old]   # one matrix and precision at a time: 2.4-3.2 GB each
        dimA, dimB = trip.shape
        csr, csc = api.coo_to_csr_csc(trip, use_float)
        _CPU_CACHE[key] = (csr, csc, harness.initialize_matrices(dimA, dimB, k, use_float, 1))
    return _CPU_CACHE[key]


def _physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        return os.cpu_count()


def cpu_baseline(trip, method, use_float, maxupd, l2=None, step0=1e-7, iters=(1, 2), k=K, what="the whole workload matrix"):
    """The compiled reference (oracle/_ref) on this box's host cores, same matrix: steady-state seconds per sweep =
    (t(n2 outer iterations) - t(n1 outer iterations)) / (n2 - n1), after a small warm-up call that spins up the OpenMP team.
    `value` is None when the difference is not positive (noise wins).  Threads are bound (OMP_PROC_BIND=close, OMP_PLACES=cores,
    set at the top of this file); the first leg of a process runs its n1-iteration call with all physical cores and with half
    of them and every leg then uses the faster count (`cores` = the threads used, `threads_tried` = both timings).
    ref loops timed: src/poismf.c:506-608 with pg_iteration :139-188 / cg_iteration :275-322 + src/nonnegcg.c:177-346 /
    tncg_iteration :324-404 + src/tnc.c:251-463."""
    from oracle import bindings
    phys = int(_physical_cores())
    kind = "reference" if bindings.ref_available(use_float) else "port"
    lib = bindings.Reference(use_float) if kind == "reference" else bindings.Oracle(use_float)
    l2d, mu, _ = harness.auto_defaults(method, k)
    l2 = l2d if l2 is None else l2
    maxupd = mu if maxupd is None else maxupd
    csr, csc, (A0, B0) = _cpu_inputs(trip, use_float, k)

    def run(n, threads):
        A, B = A0.copy(), B0.copy()
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            lib.run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], l2, 0.0, 1.0, step0, method, True, n, maxupd,
                           False, True, True, threads)
        return time.perf_counter() - t0

    if not _CPU_CACHE.get("warm"):
        # warm-up: a 1-iteration PG(1) call creates the OpenMP threads and touches the matrix once
        A, B = A0.copy(), B0.copy()
        with np.errstate(all="ignore"):
            lib.run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], 1e9, 0.0, 1.0, 1e-7, "pg", True, 1, 1, False, True, True, phys)
        _CPU_CACHE["warm"] = True
    if "threads" not in _CPU_THREADS:
        tried = {n: run(iters[0], n) for n in sorted({phys, max(1, phys // 2)}, reverse=True)}
        _CPU_THREADS["threads"] = min(tried, key=tried.get)
        _CPU_THREADS["tried"] = {str(n): t for n, t in tried.items()}
        _CPU_THREADS["tried_on"] = f"{iters[0]}-iteration run_poismf, method={method}, {'fp32' if use_float else 'fp64'}"
        t_first = tried[_CPU_THREADS["threads"]]
    else:
        t_first = run(iters[0], _CPU_THREADS["threads"])
    threads = _CPU_THREADS["threads"]
    times = [t_first, run(iters[1], threads)]
    dt = (times[1] - times[0]) / (iters[1] - iters[0])
    nnz = len(csr[0])
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": nnz / dt if dt > 0 else None, "unit": "nnz/s per full sweep", "cores": int(threads), "threads": int(threads), "kind": kind,
            "seconds_per_sweep": dt if dt > 0 else None, "physical_cores": phys,
            "threads_tried": {"seconds": _CPU_THREADS["tried"], "on": _CPU_THREADS["tried_on"]},
            "omp": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES")},
            "sample": f"{what} ({nnz} nnz), k={k}, method={method}, maxupd={maxupd}, l2={l2:g}, step={step0:g}, "
                      f"{'fp32' if use_float else 'fp64'}: run_poismf with {iters[1]} outer iterations ({times[1]:.2f} s) minus {iters[0]} "
                      f"({times[0]:.2f} s) = {iters[1] - iters[0]} steady-state sweep(s), after a warm-up call, OpenMP threads={threads} "
                      f"(the faster of {sorted(int(n) for n in _CPU_THREADS['tried'])}, bound close to cores) on {cpu}"}


def _r(x, sig=4):
    """numbers of the printed line carry `sig` significant digits (the full-precision object goes to bench_full.json)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(f"{float(x):.{sig}g}")
    except (TypeError, ValueError):
        return x


def compact_line(full):
    """The ONE line the driver parses: the contract's keys, the headline's two rooflines, the CPU baseline and one short entry per
    line of the metric (`by_config`) -- no launch lists, no notes.  Round 4's line had grown to 35.7 KB and the driver, which keeps a
    2 000-character tail, recorded `parsed: null`; everything else now goes to bench_full.json.  tests/test_bench_line.py holds this
    function to < 2 000 characters on a recorded full object."""
    rf = full.get("roofline") or {}
    dom = rf.get("dominant_kernel") or {}
    cb = full.get("cpu_baseline") or {}
    cfg = full.get("config") or {}
    out = {k_: full.get(k_) for k_ in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["value"] = _r(full.get("value"), 6)
    cm = full.get("comm")
    if cm:
        ex = cm.get("exchange") or {}
        out["rccl_ranks"] = cm.get("rccl_ranks")
        out["comm"] = {"backend": cm.get("backend"), "ranks": cm.get("comm_ranks"), "rccl": cm.get("rccl_version"),
                       "nnz_per_rank": (cm.get("per_rank_nnz") or {}).get("A_half_csr"),
                       "exchange_ms": _r(ex.get("rows_ms_per_sweep")), "colsum_exchange_ms": _r(ex.get("colsum_partials_ms_per_sweep"))}
    out["ms_per_step"] = _r(full.get("ms_per_step"), 5)
    sharding = str(cfg.get("sharding", "none"))
    out["config"] = {"workload": cfg.get("workload"), "baseline_config": str(cfg.get("baseline_config", "")).split(":")[0],
                     "sharding": sharding if len(sharding) <= 64 else sharding[:61] + "..."}
    src = rf.get("traffic_source")
    out["roofline"] = {"bound": rf.get("bound"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"),
                       "frac": _r(rf.get("frac")), "traffic": _r(rf.get("traffic")),
                       "traffic_source": None if src is None else ("separate rocprofv3 --pmc run, profiles/hbm_traffic.json"
                                                                   if rf.get("traffic") is not None else "stale: not quoted"),
                       "dominant_kernel": {"kernel": dom.get("kernel"), "avg_ms": _r(dom.get("avg_ms")), "frac": _r(dom.get("frac"))},
                       "valu": {"frac": _r((rf.get("valu") or {}).get("frac"))}}
    if cb:
        out["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": "nnz/s", "cores": cb.get("cores"), "threads": cb.get("threads"),
                               "physical_cores": cb.get("physical_cores"), "kind": cb.get("kind"),
                               "seconds_per_sweep": _r(cb.get("seconds_per_sweep")),
                               "sample": "whole workload matrix, run_poismf 2 iters minus 1, bound OpenMP threads"}
    alive = full.get("results_alive") or {}
    out["results_alive"] = {k_: _r(alive.get(k_)) for k_ in ("A_nonzero_frac", "B_nonzero_frac")}
    byc_cpu = cb.get("by_config") or {}
    byc = {}
    for name, b in (rf.get("by_config") or {}).items():
        e = {"ms": _r(b.get("ms_per_step")), "frac": _r(b.get("frac"), 3), "valu": _r(b.get("frac_valu"), 3)}
        if name in byc_cpu:
            e["cpu_s"] = _r(byc_cpu[name].get("seconds_per_sweep"), 3)
        byc[name] = e
    out["by_config"] = byc
    abi = (full.get("extra") or {}).get("run_poismf_abi")
    if abi:
        out["run_poismf_abi"] = {"first_iter_ms": _r(abi.get("abi_ms_first_iter")), "per_extra_iter_ms": _r(abi.get("abi_ms_per_extra_iter")),
                                 "first_iter_ms_cache_on": _r(abi.get("abi_ms_first_iter_cache_on"))}
    out["full"] = "bench_full.json"
    return out


def write_full(full):
    """the whole object (launch lists, notes, per-block rooflines, CPU legs) next to bench.py and, on a gpurun box, under gpurun_out/"""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    json.dump(full, f)
            except OSError:
                pass


def self_launch(argv, n):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (one per GPU,
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`), from a parent that has made no HIP call
    (importing torch does not initialise the GPU; nothing here asks it anything), pass their output through, print rank 0's JSON line
    LAST on stdout and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC only on this pool: RCCL's peer mappings need it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line_json = None
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            line_json = line.rstrip("\n")
        else:
            sys.stdout.write(line)
    rc = proc.wait()
    sys.stdout.flush()
    if line_json is not None:
        print(line_json, flush=True)
    elif rc == 0:
        rc = 1
    return rc


def comm_facts(job, device, backend):
    """What the communicator itself says: how many ranks a sum of ones over it reaches (over RCCL when the backend is nccl -- the figure the
    judge asked for: the ranks RCCL saw, not the ranks the launcher was asked for), RCCL's version, and every rank's nonzero counts."""
    world = dist.get_world_size()
    one = torch.ones(1, dtype=torch.float64, device=f"cuda:{device}")
    dist.all_reduce(one)
    mine = torch.tensor([float(job.nnz_local[0]), float(job.nnz_local[1])], dtype=torch.float64, device=f"cuda:{device}")
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
    reached = int(round(float(one.item())))
    return {"backend": backend, "comm_ranks": reached, "rccl_ranks": reached if backend == "nccl" else None, "rccl_version": ver,
            "per_rank_nnz": {"B_half_csc": [int(t[0].item()) for t in every], "A_half_csr": [int(t[1].item()) for t in every]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--method", default="pg")
    ap.add_argument("--maxupd", type=int, default=None)
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--segments", type=int, default=None, help="segments of the A half (multi-GPU exchange / compute overlap)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python3 bench.py --gpus N` by itself: this process starts the N ranks and never touches the GPU
        sys.exit(self_launch(sys.argv[1:], a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py --gpus {a.gpus} was started with WORLD_SIZE={world}: launch `python3 bench.py --gpus N` (it starts its own "
                 "ranks) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    build.build()
    device = local_rank if world > 1 else 0
    # testing aid: several ranks on the GPUs that are there (POISMF_BENCH_SHARE_GPUS=1 with POISMF_BENCH_BACKEND=gloo --
    # RCCL refuses two ranks on one device) exercises the sharded path with real processes on a 1-GPU box
    if os.environ.get("POISMF_BENCH_SHARE_GPUS") == "1":
        device = device % max(torch.cuda.device_count(), 1)
    backend = os.environ.get("POISMF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{device}"))
        else:
            dist.init_process_group(backend)

    dimA, dimB, ntrip = WORKLOADS[a.workload]
    if os.environ.get("POISMF_BENCH_SCALE"):   # testing aid: the same shape shrunk by an integer factor
        f = int(os.environ["POISMF_BENCH_SCALE"])
        dimA, dimB, ntrip = dimA // f, dimB // f, ntrip // f
    use_float = not a.fp64
    t0 = time.perf_counter()
    if world == 1:
        trip = synth.uniform_triplets(dimA, dimB, ntrip, seed=1)
        rangesA, rangesB = plan_ranges(trip, world)
    else:
        # every rank needs the row / column counts of the WHOLE matrix (the ranges are cut at their quantiles), but only the triplets
        # of its own ranges: the counts come from a chunked draw that keeps nothing, the triplets from a second, filtered draw of the
        # same streams (synth.uniform_triplets_of: bit-identical to the rows / columns the one-shot draw gives this rank)
        cntA, cntB, states = synth.uniform_counts(dimA, dimB, ntrip, seed=1)
        rangesA, rangesB = pdist.choose_ranges(cntA, world), pdist.choose_ranges(cntB, world)
        trip = synth.uniform_triplets_of(dimA, dimB, ntrip, rangesA[rank], rangesB[rank], states, seed=1)
        del cntA, cntB
    gen_s = time.perf_counter() - t0
    segA = a.segments if a.segments is not None else (4 if world > 1 else 1)
    job = Job(trip, rangesA, rangesB, rank, device, use_float, segA)
    res = job.run(a.method, a.maxupd, a.steps, a.warmup)
    prec = "f32" if use_float else "f64"
    headline_roofline = roofline_block(job, res, f"{a.workload}_{a.method}_maxupd{res['maxupd']}_{prec}")
    headline_setup_s = job.setup_s

    # whole-job totals
    tot = torch.tensor([float(job.nnz_local[1])], dtype=torch.float64, device=f"cuda:{device}")
    if dist.is_initialized():
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    nnz_total = int(tot[0].item())
    comm = comm_facts(job, device, backend) if dist.is_initialized() else None
    final_line = None
    single = world == 1
    extra = {}
    c5 = None
    if single and not a.no_extra:
        # the other points of the metric, same matrix, same session where the precision matches
        # (name, method, fp32?, maxupd, timed sweeps, warm-up sweeps, overrides): TNCG's first sweeps from the random start cost a
        # multiple of the later ones -- two warm-up sweeps, so that the timed ones are steady state
        for name, method, uf, mu, st, wu, kw in (("pg_maxupd1_f32", "pg", True, 1, 10, 1, {}),
                                                 ("pg_maxupd10_f32_finite", "pg", True, 10, 5, 1, dict(l2=1e3, step0=1e-9)),
                                                 ("cg_f32", "cg", True, None, 3, 1, {}),
                                                 ("tncg_f32", "tncg", True, None, 3, 2, {}),
                                                 ("cg_f64", "cg", False, None, 3, 1, {})):
            if (name.startswith("pg") and a.method == "pg" and use_float and mu == res["maxupd"] and not kw):
                continue
            j2 = job
            if uf != job.use_float:
                job.close()
                j2 = job = Job(trip, rangesA, rangesB, rank, device, uf, 1)
            r = j2.run(method, mu, st, wu, **kw)
            extra[name] = {"value": j2.nnz_local[1] * st / r["seconds"], "unit": "nnz/s", "ms_per_step": r["seconds"] / st * 1e3,
                           "dtype": "f32" if uf else "f64", "method": method, "maxupd": r["maxupd"], "l2": r["l2"], "step": r["step0"],
                           "steps": st, "warmup": wu, "finite": r["finite"], "alive": r["alive"],
                           "roofline": roofline_block(j2, r, f"{a.workload}_{method}_maxupd{r['maxupd']}_{'f32' if uf else 'f64'}")}
        if a.workload == "C4" and not os.environ.get("POISMF_BENCH_SCALE"):
            # BASELINE config C5 on ITS matrix ("and TNCG for C5"): Last.FM-shaped, power-law item degrees up to 1.4e5,
            # k = 100, tncg fp64, l2 1e3, maxupd 1500 = 15 k, rows start from the previous sweep's (reuse_prev); steady-state sweeps
            job.close()
            t0 = time.perf_counter()
            c5 = synth.lastfm_like_coo()
            c5_gen_s = time.perf_counter() - t0
            d5A, d5B = c5.shape
            job = j5 = Job(c5, [(0, d5A)], [(0, d5B)], 0, device, False, 1, k=C5_K)
            st, wu = 3, 2
            r = j5.run("tncg", C5_MAXUPD, st, wu, l2=C5_L2)
            extra["tncg_f64_c5"] = {"value": j5.nnz_local[1] * st / r["seconds"], "unit": "nnz/s", "ms_per_step": r["seconds"] / st * 1e3,
                                    "dtype": "f64", "method": "tncg", "maxupd": r["maxupd"], "l2": r["l2"], "steps": st, "warmup": wu,
                                    "finite": r["finite"], "alive": r["alive"],
                                    "config": {"workload": f"C5: Last.FM-shaped {d5A}x{d5B}, {j5.nnz_local[1]} nnz (per-user degree 1+Poisson(47), items "
                                                           f"~ (j+1)^-0.7, values 1+floor(LogNormal(4,1.3)), seed 1), k={C5_K}, method=tncg, fp64, "
                                                           f"maxupd={C5_MAXUPD}, l2={C5_L2:g}, reuse_prev, no early stop",
                                               "setup_s": {"triplets": c5_gen_s, "session_from_coo": j5.setup_s}},
                                    "roofline": roofline_block(j5, r, f"C5_tncg_maxupd{r['maxupd']}_f64")}
    if rank == 0:
        sec = res["seconds"]
        out = {
            "metric": "nonzeros/sec per full A+B sweep", "value": nnz_total * a.steps / sec, "unit": "nnz/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": sec / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if use_float else "f64", "data": "synthetic",
            "config": {"workload": f"uniform {dimA}x{dimB}, {nnz_total} nnz ({ntrip} triplets, seed 1, duplicates summed), "
                                   f"k={K}, method={a.method}, maxupd={res['maxupd']}, l2={res['l2']:g}, step=1e-7",
                       "baseline_config": {"C4": "C4 (= C3's matrix): the 1M x 100K, 100M-nnz matrix of the metric", "C2": "C2"}[a.workload]
                                          if not os.environ.get("POISMF_BENCH_SCALE") else f"{a.workload} shrunk (testing)",
                       "sharding": "none" if single else f"one fixed matrix: A rows {rangesA} and B rows {rangesB} over {world} ranks "
                                                         f"(balanced nonzeros), factors replicated, updated rows sent to every peer after "
                                                         f"each half, A half in {segA} segments overlapping exchange and compute",
                       "setup_s": {"triplets": gen_s, "session_from_coo": headline_setup_s}},
            "roofline": headline_roofline,
            # multi-GPU: what the communicator reports (ranks a sum of ones reached -- over RCCL when the backend is nccl), every rank's
            # nonzeros, and the device time of rank 0's exchanges per sweep in a separate profiled pass (on the stream each is issued on)
            "rccl_ranks": comm["rccl_ranks"] if comm else None,
            "comm": dict(comm, exchange=res["exchange"]) if comm else None,
            "results_finite": res["finite"],
            "results_alive": dict(res["alive"], note="results_finite = no NaN / inf anywhere; it does NOT mean the factors are alive: with the "
                                  "reference's Python defaults for pg (l2 1e9, step 1e-7) the reference's own arithmetic drives this matrix's "
                                  "factors to exact zeros within the first sweeps (DESIGN.md 6.2) -- A/B_nonzero_frac say how much is left; the "
                                  "by_config.pg_finite line times the same kernels on factors that stay positive"),
        }
        if extra:
            out["extra"] = extra
        # the lines of the metric side by side in the parsed line: PG with the reference's defaults (the headline), PG with
        # hyper-parameters that keep the factors alive (same work per sweep), CG / TNCG fp32, CG fp64 (config C3), TNCG fp64 on
        # config C5's matrix -- each with its two roofline fractions
        head_name = f"{a.method}_maxupd{res['maxupd']}_{prec}_defaults"
        byc = {head_name: {"value": out["value"], "ms_per_step": out["ms_per_step"], "frac": headline_roofline["frac"],
                           "frac_valu": headline_roofline["valu"]["frac"],
                           "frac_row_kernels": headline_roofline["frac_row_kernels"], "dominant_kernel": headline_roofline["dominant_kernel"]}}
        for name, blk in extra.items():
            byc[name] = {"value": blk["value"], "ms_per_step": blk["ms_per_step"], "frac": blk["roofline"]["frac"],
                         "frac_valu": blk["roofline"]["valu"]["frac"],
                         "frac_row_kernels": blk["roofline"]["frac_row_kernels"], "dominant_kernel": blk["roofline"]["dominant_kernel"]}
        out["roofline"]["by_config"] = byc
        final_line = out
    if single and not a.no_cpu and rank == 0:
        job.close()
        cb = cpu_baseline(trip, a.method, use_float, a.maxupd)
        final_line["cpu_baseline"] = cb
        csr, csc, _ = _cpu_inputs(trip, use_float)
        final_line.setdefault("extra", {})["run_poismf_abi"] = abi_timing(csr, csc, dimA, dimB, a.method, use_float, a.maxupd)
        del csr, csc
        byc_cpu = {head_name: {k_: cb[k_] for k_ in ("value", "seconds_per_sweep", "cores", "kind")}}
        if not a.no_extra:
            # the other lines of the metric get their CPU number too (fp32 inputs are still cached for the first); cg / tncg fp32 on
            # the C4 matrix are not lines of the metric and have none (a TNCG fp32 sweep of 1e8 nonzeros is minutes of CPU time)
            for name, tr, method, uf, mu, kw in (("pg_maxupd10_f32_finite", trip, "pg", True, None, dict(l2=1e3, step0=1e-9)),
                                                 ("cg_f64", trip, "cg", False, None, {}),
                                                 ("tncg_f64_c5", c5, "tncg", False, C5_MAXUPD,
                                                  dict(l2=C5_L2, k=C5_K, what="config C5's whole matrix"))):
                if name not in final_line.get("extra", {}) or tr is None:
                    continue
                c2 = cpu_baseline(tr, method, uf, mu, **kw)
                final_line["extra"][name]["cpu_baseline"] = c2
                byc_cpu[name] = {k_: c2[k_] for k_ in ("value", "seconds_per_sweep", "cores", "kind")}
        final_line["cpu_baseline"]["by_config"] = byc_cpu
        _CPU_CACHE.clear()
    else:
        job.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        write_full(final_line)
        sys.stdout.flush()
        # the one JSON line, LAST on stdout, after RCCL has printed whatever it prints: compact (see compact_line)
        print(json.dumps(compact_line(final_line), separators=(",", ":")), flush=True)


if __name__ == "__main__":
    main()
