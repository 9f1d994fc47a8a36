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
default), pg with hyper-parameters that keep the factors finite (same work per sweep), and cg fp64 (config C3).
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

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")  # the CPU baseline threads over rows with OpenMP, as the reference does
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from poismf_amd import api, build, harness, synth  # noqa: E402
from poismf_amd import dist as pdist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOADS = {
    # name: (dimA, dimB, triplets)
    "C4": (10 ** 6, 10 ** 5, 10 ** 8),   # = C3's matrix; the one BASELINE.json's metric is quoted on
    "C2": (10 ** 5, 10 ** 5, 10 ** 7),   # development only
}
K = 50


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

    def __init__(self, trip, rangesA, rangesB, rank, device, use_float, segmentsA):
        self.trip, self.rangesA, self.rangesB, self.rank, self.device, self.use_float = trip, rangesA, rangesB, rank, device, use_float
        self.dimA, self.dimB = trip.shape
        t0 = time.perf_counter()
        self.be = pdist.HipBackend(None, None, self.dimA, self.dimB, K, use_float,
                                   dict(method="pg", l2_reg=1.0), rangesA[rank], rangesB[rank], device, coo=trip,
                                   segments=(1, segmentsA))
        torch.cuda.synchronize(device)
        self.setup_s = time.perf_counter() - t0
        self.A0, self.B0 = harness.initialize_matrices(self.dimA, self.dimB, K, use_float, 1)
        self.nnz_local = (self.be.sess.nnz(0), self.be.sess.nnz(1))

    def run(self, method, maxupd, steps, warmup, l2=None, step0=1e-7, profile_steps=None):
        """warmup + `steps` timed sweeps with profiling OFF (the product configuration), then a separate profiled pass
        (HIP events around the row-kernel launches of each half, per-row pass counters) of `profile_steps` sweeps from
        the same starting point for the kernel time and the pass-weighted traffic."""
        l2d, mu, _ = harness.auto_defaults(method, K)
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
        for _ in range(psteps):
            alt.sweep()
        k_ms = [sess.kernel_time(w) for w in (0, 1)]
        ev_stats = [sess.eval_stats(w) for w in (0, 1)]
        plan = [sess.plan(w) for w in (0, 1)]
        lprof = [sess.launch_profile(w) for w in (0, 1)]
        sess.profile(False)
        return dict(method=method, maxupd=maxupd, l2=l2, step0=step0, steps=steps, seconds=dt, finite=finite, alive=alive, psteps=psteps,
                    kernel_ms=k_ms, ev_stats=ev_stats, plan=plan, lprof=lprof)

    def close(self):
        self.be.close()


def roofline_block(job, res, traffic_key=None):
    """HBM roofline of one run, this rank's launches.  `achieved` / `frac`: SURVEY.md 8(d)'s algorithmic bytes of the rank's two
    shards per sweep / the UNPROFILED time of a sweep (the timed region: row kernels + column sums, profiling off).  The
    row-kernel durations (`kernel_ms_*`, `frac_row_kernels`, `launches`) come from a separate profiled pass: HIP events on the
    stream each launch is issued on, around every row-bin launch and around each half."""
    s = 4 if job.use_float else 8
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
    traffic = None
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if traffic_key and os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(traffic_key)
        except Exception:
            traffic = None
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
                    "bytes = nnz*(4+s+k*s) + 2*dimM*k*s + (dimM+1)*8 per half; traffic = fabric-side bytes per sweep from separate PMC passes"}


def abi_timing(csr, csc, dimA, dimB, method, use_float, maxupd):
    """Wall time of the drop-in run_poismf() itself -- host arrays in, host arrays out, set-up (upload, index narrowing and
    row sort on the device) included: the PCIe-inclusive cost a caller of the reference's ABI sees.  Never `value`."""
    l2, mu, _ = harness.auto_defaults(method, K)
    maxupd = mu if maxupd is None else maxupd
    A0, B0 = harness.initialize_matrices(dimA, dimB, K, use_float, 1)
    t = {}
    for numiter in (1, 1, 6):
        A, B = A0.copy(), B0.copy()
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            api._run_poismf(csr[0], csr[1], csr[2], csc[0], csc[1], csc[2], A, B, method, True, l2, 0., 1., 1e-7, numiter, maxupd,
                            False, True, True, 1)
        t[numiter] = (time.perf_counter() - t0) * 1e3
    return {"abi_ms_first_iter": t[1], "abi_ms_per_extra_iter": (t[6] - t[1]) / 5.0,
            "note": f"run_poismf(method={method}, maxupd={maxupd}, {'fp32' if use_float else 'fp64'}) on the workload matrix through "
                    "ctypes, second call of the process (the first also pays device initialisation)"}


_CPU_CACHE = {}


def _cpu_inputs(trip, use_float):
    """host CSR / CSC (size_t indices) for the reference's ABI + the starting factors, once per precision"""
    if use_float not in _CPU_CACHE:
        _CPU_CACHE.clear()   # one precision at a time: 2.4-3.2 GB each
        dimA, dimB = trip.shape
        csr, csc = api.coo_to_csr_csc(trip, use_float)
        _CPU_CACHE[use_float] = (csr, csc, harness.initialize_matrices(dimA, dimB, K, use_float, 1))
    return _CPU_CACHE[use_float]


def cpu_baseline(trip, method, use_float, maxupd, l2=None, step0=1e-7, iters=(1, 2)):
    """The compiled reference (oracle/_ref) on this box's host cores, same matrix: steady-state seconds per sweep =
    (t(n2 outer iterations) - t(n1 outer iterations)) / (n2 - n1), after a small warm-up call that spins up the OpenMP team.
    `value` is None when the difference is not positive (noise wins).  ref loops timed: src/poismf.c:506-608 with
    pg_iteration :139-188 / cg_iteration :275-322 + src/nonnegcg.c:177-346."""
    from oracle import bindings
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        cores = os.cpu_count()
    kind = "reference" if bindings.ref_available(use_float) else "port"
    lib = bindings.Reference(use_float) if kind == "reference" else bindings.Oracle(use_float)
    l2d, mu, _ = harness.auto_defaults(method, K)
    l2 = l2d if l2 is None else l2
    maxupd = mu if maxupd is None else maxupd
    csr, csc, (A0, B0) = _cpu_inputs(trip, use_float)

    def run(n, rows=None):
        A, B = A0.copy(), B0.copy()
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            lib.run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], l2, 0.0, 1.0, step0, method, True, n, maxupd,
                           False, True, True, cores)
        return time.perf_counter() - t0

    if not _CPU_CACHE.get("warm"):
        # warm-up: a 1-iteration PG(1) call creates the OpenMP threads and touches the matrix once
        A, B = A0.copy(), B0.copy()
        with np.errstate(all="ignore"):
            lib.run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], 1e9, 0.0, 1.0, 1e-7, "pg", True, 1, 1, False, True, True, cores)
        _CPU_CACHE["warm"] = True
    times = [run(n) for n in iters]
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
    return {"value": nnz / dt if dt > 0 else None, "unit": "nnz/s per full sweep", "cores": int(cores), "kind": kind,
            "seconds_per_sweep": dt if dt > 0 else None,
            "sample": f"the whole workload matrix ({nnz} nnz), method={method}, maxupd={maxupd}, l2={l2:g}, step={step0:g}, "
                      f"{'fp32' if use_float else 'fp64'}: run_poismf with {iters[1]} outer iterations ({times[1]:.2f} s) minus {iters[0]} "
                      f"({times[0]:.2f} s) = {iters[1] - iters[0]} steady-state sweep(s), after a warm-up call, OpenMP threads={cores} on {cpu}"}


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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world == 1 and a.gpus > 1:
        sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
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
    trip = synth.uniform_triplets(dimA, dimB, ntrip, seed=1)
    gen_s = time.perf_counter() - t0
    rangesA, rangesB = plan_ranges(trip, world)
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
    final_line = None
    single = world == 1
    extra = {}
    if single and not a.no_extra:
        # the other points of the metric, same matrix, same session where the precision matches
        for name, method, uf, mu, st, kw in (("pg_maxupd1_f32", "pg", True, 1, 10, {}),
                                             ("pg_maxupd10_f32_finite", "pg", True, 10, 5, dict(l2=1e3, step0=1e-9)),
                                             ("cg_f64", "cg", False, None, 3, {})):
            if (name.startswith("pg") and a.method == "pg" and use_float and mu == res["maxupd"] and not kw):
                continue
            j2 = job
            if uf != job.use_float:
                job.close()
                j2 = job = Job(trip, rangesA, rangesB, rank, device, uf, 1)
            r = j2.run(method, mu, st, 1, **kw)
            extra[name] = {"value": j2.nnz_local[1] * st / r["seconds"], "unit": "nnz/s", "ms_per_step": r["seconds"] / st * 1e3,
                           "dtype": "f32" if uf else "f64", "method": method, "maxupd": r["maxupd"], "l2": r["l2"], "step": r["step0"],
                           "steps": st, "finite": r["finite"], "alive": r["alive"],
                           "roofline": roofline_block(j2, r, f"{a.workload}_{method}_maxupd{r['maxupd']}_{'f32' if uf else 'f64'}")}
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
            "results_finite": res["finite"],
            "results_alive": dict(res["alive"], note="results_finite = no NaN / inf anywhere; it does NOT mean the factors are alive: with the "
                                  "reference's Python defaults for pg (l2 1e9, step 1e-7) the reference's own arithmetic drives this matrix's "
                                  "factors to exact zeros within the first sweeps (DESIGN.md 6.1) -- A/B_nonzero_frac say how much is left; the "
                                  "by_config.pg_finite line times the same kernels on factors that stay positive"),
        }
        if extra:
            out["extra"] = extra
        # the three lines of the metric side by side in the parsed line: PG with the reference's defaults (the headline), PG with
        # hyper-parameters that keep the factors alive (same work per sweep), CG fp64 (config C3) -- each with its roofline numbers
        head_name = f"{a.method}_maxupd{res['maxupd']}_{prec}_defaults"
        byc = {head_name: {"value": out["value"], "ms_per_step": out["ms_per_step"], "frac": headline_roofline["frac"],
                           "frac_row_kernels": headline_roofline["frac_row_kernels"], "dominant_kernel": headline_roofline["dominant_kernel"]}}
        for name, blk in extra.items():
            byc[name] = {"value": blk["value"], "ms_per_step": blk["ms_per_step"], "frac": blk["roofline"]["frac"],
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
            # the other two lines of the metric get their CPU number too (fp32 inputs are still cached for the first)
            for name, method, uf, kw in (("pg_maxupd10_f32_finite", "pg", True, dict(l2=1e3, step0=1e-9)), ("cg_f64", "cg", False, {})):
                if name not in final_line.get("extra", {}):
                    continue
                c2 = cpu_baseline(trip, method, uf, None, **kw)
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
        sys.stdout.flush()
        print(json.dumps(final_line), flush=True)  # the one JSON line, after RCCL has printed whatever it prints


if __name__ == "__main__":
    main()
