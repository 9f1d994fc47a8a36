#!/usr/bin/env python3
"""bench.py -- nonzeros/sec per full A+B sweep of the factor-update hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--method pg|cg|tncg] [--maxupd M] [--no-cpu] [--no-extra]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full outer iteration (B half + A half, column sums and -- multi-GPU -- the shard
all-gathers included) on synthetic data that is already resident in HBM when the timed region starts.

N = 1  : BASELINE config C2 -- uniform 1e5 x 1e5, 1e7 triplets (9 994 947 nnz after duplicate summing),
         k = 50, method = pg, fp32, the reference's Python defaults (l2 1e9, step 1e-7, maxupd 10).
N > 1  : weak scaling of that shape: every rank owns one 1e5-row block of A (1e7 triplets, seed 1 + rank),
         so X is (N*1e5) x 1e5 with ~N*1e7 nnz -- N = 8 is roughly BASELINE config C4.  B rows are split
         evenly; after each half the updated shard is all-gathered over RCCL.

Rank 0 prints ONE JSON line (see the driver contract); `roofline` prices the row-update kernel against
HBM peak with SURVEY.md 8(d)'s algorithmic bytes, `cpu_baseline` is the compiled reference (oracle/_ref,
kind "reference") timed on this box's host cores on the same matrix.
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
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from poismf_amd import api, build, harness, synth  # noqa: E402
from poismf_amd import dist as pdist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BLOCK_ROWS, DIMB, BLOCK_NNZ, K = 10 ** 5, 10 ** 5, 10 ** 7, 50


def algorithmic_bytes_half(nnz, dimM, k, s):
    """one gather per stored nonzero; device widths s_idx = 4, s_ptr = 8."""
    return nnz * (4 + s + k * s) + 2 * dimM * k * s + (dimM + 1) * 8


def build_inputs(rank, world, use_float, BLOCK_ROWS=BLOCK_ROWS, DIMB=DIMB, BLOCK_NNZ=BLOCK_NNZ):
    """Returns (csr, csc, dimA, dimB, rangesA, rangesB) with only this rank's shards populated in the
    whole-matrix-shaped CSR / CSC arrays the C-ABI takes (tests/test_bench_inputs.py checks that the shards of all
    ranks tile the CSR / CSC of the stacked blocks exactly)."""
    dt = np.float32 if use_float else np.float64
    dimA, dimB = BLOCK_ROWS * world, DIMB
    rangesA = [(r * BLOCK_ROWS, (r + 1) * BLOCK_ROWS) for r in range(world)]
    rangesB = pdist.equal_ranges(dimB, world)
    if world == 1:
        coo = synth.uniform_coo(dimA, dimB, BLOCK_NNZ, seed=1)
        csr, csc = harness.process_data(coo, use_float)
        return csr, csc, dimA, dimB, rangesA, rangesB
    # CSR: own block only, placed at its global row offset
    own = sp.csr_matrix(synth.uniform_coo(BLOCK_ROWS, dimB, BLOCK_NNZ, seed=1 + rank))
    own.sum_duplicates(); own.sort_indices()
    ptr = np.zeros(dimA + 1, np.uint64)
    r0 = rank * BLOCK_ROWS
    ptr[r0:r0 + BLOCK_ROWS + 1] = own.indptr
    ptr[r0 + BLOCK_ROWS + 1:] = own.indptr[-1]
    csr = (own.data.astype(dt), own.indices.astype(np.uint64), ptr)
    # CSC: this rank's column range of EVERY block
    c0, c1 = rangesB[rank]
    rows, cols, vals = [], [], []
    for r in range(world):
        blk = synth.uniform_coo(BLOCK_ROWS, dimB, BLOCK_NNZ, seed=1 + r)
        m = (blk.col >= c0) & (blk.col < c1)
        rows.append(blk.row[m] + r * BLOCK_ROWS); cols.append(blk.col[m] - c0); vals.append(blk.data[m])
    sub = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(dimA, c1 - c0))
    sub.sum_duplicates(); sub.sort_indices()
    cptr = np.zeros(dimB + 1, np.uint64)
    cptr[c0:c1 + 1] = sub.indptr
    cptr[c1 + 1:] = sub.indptr[-1]
    csc = (sub.data.astype(dt), sub.indices.astype(np.uint64), cptr)
    return csr, csc, dimA, dimB, rangesA, rangesB


def timed_sweeps(alt, steps, warmup, world, device):
    multi = dist.is_initialized()
    for _ in range(warmup):
        alt.sweep()
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


def run_gpu(csr, csc, dimA, dimB, rangesA, rangesB, rank, world, device, method, use_float, maxupd, steps, warmup, seed=1):
    l2, mu, _ = harness.auto_defaults(method, K)
    maxupd = mu if maxupd is None else maxupd
    A0, B0 = harness.initialize_matrices(dimA, dimB, K, use_float, seed)
    be = pdist.HipBackend(csr, csc, dimA, dimB, K, use_float,
                          dict(method=method, l2_reg=l2, maxupd=maxupd, limit_step=True, early_stop=False, reuse_prev=True),
                          rangesA[rank], rangesB[rank], device)
    be.sess.set_factors(A0, B0)
    alt = pdist.ShardedAlternation(be, rangesA, rangesB, method, l2, 1e-7, dims=(dimA, dimB))
    for _ in range(warmup):
        alt.sweep()
    be.sess.profile(True)
    dt = timed_sweeps(alt, steps, 0, world, device)
    k_ms = [be.sess.kernel_time(w) for w in (0, 1)]
    ev_stats = [be.sess.eval_stats(w) for w in (0, 1)]   # (tile passes, passes x nonzeros) of the timed sweeps
    nnz_local = (be.sess.nnz(0), be.sess.nnz(1))
    A, B = be.sess.get_factors()
    be.close()
    return dict(seconds=dt, kernel_ms=k_ms, ev_stats=ev_stats, nnz_local=nnz_local, finite=bool(np.isfinite(A).all() and np.isfinite(B).all()),
                maxupd=maxupd, l2=l2)


def pass_weighted(res, rows, steps, k, s):
    """what the inner solvers actually read from the on-chip tiles -- sum over rows of (passes over
    the row's tile) x nonzeros x k x sizeof, per sweep, from counters the row kernels keep while profiling is on."""
    passes = res["ev_stats"][0][0] + res["ev_stats"][1][0]
    nnzp = res["ev_stats"][0][1] + res["ev_stats"][1][1]
    k_ms = (res["kernel_ms"][0][0] + res["kernel_ms"][1][0]) / steps
    gb = nnzp * k * s / steps / 1e9
    return {"tile_passes_per_row": passes / steps / max(rows, 1), "on_chip_GB_per_sweep": gb,
            "on_chip_GBps": gb / (k_ms * 1e-3) if k_ms > 0 else 0.0}


def cpu_baseline(csr, csc, dimA, dimB, method, use_float, maxupd):
    """The compiled reference (oracle/_ref) on this box's host cores, same matrix, 2 full sweeps."""
    from oracle import bindings
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        cores = os.cpu_count()
    kind = "reference" if bindings.ref_available(use_float) else "port"
    lib = bindings.Reference(use_float) if kind == "reference" else bindings.Oracle(use_float)
    l2, mu, _ = harness.auto_defaults(method, K)
    maxupd = mu if maxupd is None else maxupd
    A, B = harness.initialize_matrices(dimA, dimB, K, use_float, 1)
    sweeps = 2
    t0 = time.perf_counter()
    lib.run_poismf(A, csr[0], csr[2], csr[1], B, csc[0], csc[2], csc[1], l2, 0.0, 1.0, 1e-7, method, True, sweeps, maxupd,
                   False, True, True, cores)
    dt = time.perf_counter() - t0
    nnz = len(csr[0])
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": nnz * sweeps / dt, "unit": "nnz/s per full sweep", "cores": int(cores), "kind": kind,
            "sample": f"{sweeps} full A+B sweeps of the whole workload matrix ({nnz} nnz), method={method}, maxupd={maxupd}, "
                      f"{'fp32' if use_float else 'fp64'}, OpenMP threads={cores} on {cpu}, {dt:.2f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--method", default="pg")
    ap.add_argument("--maxupd", type=int, default=None)
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--k", type=int, default=None, help="factor dimension (default 50: the BASELINE configs)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    a = ap.parse_args()
    global K
    if a.k:
        K = a.k

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
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
    force_dist = os.environ.get("POISMF_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path on a single GPU (testing)
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_dist and "RANK" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(f"cuda:{device}"))
        else:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{device}"))
            else:
                dist.init_process_group(backend)

    use_float = not a.fp64
    s = 4 if use_float else 8
    csr, csc, dimA, dimB, rangesA, rangesB = build_inputs(rank, world, use_float)
    res = run_gpu(csr, csc, dimA, dimB, rangesA, rangesB, rank, world, device, a.method, use_float, a.maxupd, a.steps, a.warmup)

    # whole-job totals
    nnz_csr_local = res["nnz_local"][1]
    tot = torch.tensor([float(nnz_csr_local), res["kernel_ms"][0][0], res["kernel_ms"][1][0]], dtype=torch.float64,
                       device=f"cuda:{device}")
    kmax = tot.clone()
    if dist.is_initialized():
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    nnz_total = int(tot[0].item())
    final_line = None
    if rank == 0:
        sec = res["seconds"]
        # roofline of the dominant kernel (half_sweep_kernel), this rank's launches: algorithmic bytes of the
        # rank's two shards per sweep / its kernel time per sweep
        b_half = [algorithmic_bytes_half(res["nnz_local"][0], rangesB[0][1] - rangesB[0][0], K, s),
                  algorithmic_bytes_half(res["nnz_local"][1], rangesA[0][1] - rangesA[0][0], K, s)]
        k_ms_sweep = (res["kernel_ms"][0][0] + res["kernel_ms"][1][0]) / a.steps
        launches = res["kernel_ms"][0][1] + res["kernel_ms"][1][1]
        achieved = sum(b_half) / (k_ms_sweep * 1e-3) / 1e9 if k_ms_sweep > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(f"{a.method}_maxupd{res['maxupd']}_{'f32' if use_float else 'f64'}")
            except Exception:
                traffic = None
        out = {
            "metric": "nonzeros/sec per full A+B sweep", "value": nnz_total * a.steps / sec, "unit": "nnz/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": sec / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if use_float else "f64", "data": "synthetic",
            "config": {"workload": f"uniform {dimA}x{dimB}, {nnz_total} nnz (1e7 triplets per 1e5-row block, duplicates summed), "
                                   f"k={K}, method={a.method}, maxupd={res['maxupd']}, l2={res['l2']:g}, step=1e-7",
                       "baseline_config": "C2" if world == 1 else f"C2 x {world} row blocks (weak scaling towards C4)",
                       "sharding": "none" if world == 1 else f"A rows and B rows split over {world} ranks, factors replicated, "
                                                              "RCCL all-gather of the updated shard after each half"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "half_sweep_reg_kernel<float, pg, S> (register-tile row kernel; one launch per row-length bin, "
                                   "S = 24/28/32/40 tile steps on this workload)",
                         "kernel_ms_per_sweep": k_ms_sweep,
                         "algorithmic_bytes_per_sweep": int(sum(b_half)), "half_sweeps_timed": int(launches),
                         "note": "achieved = algorithmic bytes of one sweep's row-kernel launches / their summed duration (HIP events on "
                                 "the session stream around each half's launches; serial launches, so this equals sum(Calls x AverageNs) "
                                 "of the half_sweep_* rows of profiles/r01/kt_pg10_kernel_stats.csv); bytes = nnz*(4+s+k*s) + 2*dimM*k*s + "
                                 "(dimM+1)*8 per half; traffic = fabric-side bytes per sweep from the PMC passes in profiles/r01/"},
            "results_finite": res["finite"],
        }
        out["roofline"]["pass_weighted"] = pass_weighted(res, (rangesA[0][1] - rangesA[0][0]) + (rangesB[0][1] - rangesB[0][0]), a.steps, K, s)
        if world == 1 and not a.no_extra:
            extra = {}
            for name, method, uf, mu, st in (("pg_maxupd1_f32", "pg", True, 1, 10), ("cg_f64", "cg", False, None, 3)):
                c2, cc2 = (csr, csc) if uf == use_float else harness.process_data(
                    sp.coo_matrix(synth.uniform_coo(dimA, dimB, BLOCK_NNZ, seed=1)), uf)
                r = run_gpu(c2, cc2, dimA, dimB, rangesA, rangesB, 0, 1, device, method, uf, mu, st, 2)
                ss = 4 if uf else 8
                bb = algorithmic_bytes_half(r["nnz_local"][0], dimB, K, ss) + algorithmic_bytes_half(r["nnz_local"][1], dimA, K, ss)
                km = (r["kernel_ms"][0][0] + r["kernel_ms"][1][0]) / st
                extra[name] = {"value": r["nnz_local"][1] * st / r["seconds"], "unit": "nnz/s", "ms_per_step": r["seconds"] / st * 1e3,
                               "roofline_frac": bb / (km * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms_per_sweep": km,
                               "finite": r["finite"], "pass_weighted": pass_weighted(r, dimA + dimB, st, K, ss)}
            out["extra"] = extra
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(csr, csc, dimA, dimB, a.method, use_float, a.maxupd)
        final_line = json.dumps(out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(final_line, flush=True)  # the one JSON line, after RCCL has printed whatever it prints


if __name__ == "__main__":
    main()
