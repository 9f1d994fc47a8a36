#!/usr/bin/env python3
"""Run one BASELINE config (C2..C5) on the GPU through the session API: timing per sweep, roofline fraction,
and a sampled-row parity check of the last half-sweep against the oracle.

    python scripts/run_config.py C3 [--method cg] [--fp32|--fp64] [--maxupd N] [--sweeps 3] [--warmup 1]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import bindings  # noqa: E402
from poismf_amd import api, harness, synth  # noqa: E402


def sub_csr(data, indices, indptr, rows):
    ip = indptr.astype(np.int64)
    lens = ip[rows + 1] - ip[rows]
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    idx = np.concatenate([np.arange(ip[r], ip[r + 1]) for r in rows]) if len(rows) else np.zeros(0, np.int64)
    return np.ascontiguousarray(data[idx]), np.ascontiguousarray(indices[idx]), ptr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--method", default=None)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--maxupd", type=int, default=None)
    ap.add_argument("--sweeps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sample", type=int, default=200)
    a = ap.parse_args()
    builder, k, method, use_float = synth.CONFIGS[a.config]
    method = a.method or method
    if a.fp32: use_float = True
    if a.fp64: use_float = False
    t0 = time.time()
    coo = builder()
    csr, csc = harness.process_data(coo, use_float)
    dimA, dimB = coo.shape
    del coo
    nnz = len(csr[0])
    t_data = time.time() - t0
    l2, maxupd, _ = harness.auto_defaults(method, k)
    if a.maxupd is not None: maxupd = a.maxupd
    A0, B0 = harness.initialize_matrices(dimA, dimB, k, use_float, 1)
    t0 = time.time()
    s = api.Session(csr, csc, dimA, dimB, k, use_float)
    s.set_factors(A0, B0)
    t_up = time.time() - t0
    p = s.make_params(method, l2, maxupd=maxupd, limit_step=True, early_stop=False, reuse_prev=(method == "tncg"))
    step = 1e-7
    for _ in range(a.warmup):
        step = s.sweep(p, step)
    s.kernel_time(0)  # synchronises the session stream
    if not os.environ.get("RUN_CONFIG_NO_PROFILE"):
        s.profile(True)
    t0 = time.time()
    for _ in range(a.sweeps - 0):
        if _ == a.sweeps - 1:
            s.kernel_time(0); tl = time.time()
            prevA, prevB = s.get_factors()
            tl = time.time() - tl
            t0 += tl
            step_last = step
        step = s.sweep(p, step)
    s.kernel_time(0)
    dt = (time.time() - t0) / a.sweeps
    kms = [s.kernel_time(w) for w in (0, 1)]
    A1, B1 = s.get_factors()
    launches = []
    if not os.environ.get("RUN_CONFIG_NO_PROFILE"):
        for w in (0, 1):
            for L in s.launch_profile(w):
                launches.append(dict(half="A" if w else "B", kernel=L["kernel"], rows=L["rows"], nnz=L["nnz"], ms_per_call=L["ms"] / max(L["calls"], 1)))
    sz = 4 if use_float else 8
    bytes_sweep = sum(n * (4 + sz + k * sz) + 2 * d * k * sz + (d + 1) * 8 for n, d in ((nnz, dimA), (nnz, dimB)))
    kern = (kms[0][0] + kms[1][0]) / a.sweeps
    out = dict(config=a.config, method=method, dtype="f32" if use_float else "f64", k=k, dimA=dimA, dimB=dimB, nnz=nnz,
               maxupd=maxupd, ms_per_sweep=dt * 1e3, nnz_per_s=nnz / dt, kernel_ms_per_sweep=kern,
               kernel_ms_B_half=kms[0][0] / a.sweeps, kernel_ms_A_half=kms[1][0] / a.sweeps,
               roofline_frac=bytes_sweep / (kern * 1e-3) / 8e12, data_build_s=t_data, upload_s=t_up,
               finite=bool(np.isfinite(A1).all() and np.isfinite(B1).all()), launches=launches)
    # sampled-row parity of the LAST sweep's A half (it used B1 as the fixed factor; its input rows are unknown
    # for CG/TNCG after the B half only changed B, so the A rows entering the A half are prevA)
    if a.sample > 0:
        orc = bindings.Oracle(use_float)
        rng = np.random.default_rng(3)
        rows = np.sort(rng.choice(dimA, min(a.sample, dimA), replace=False))
        sd, si, sp = sub_csr(csr[0], csr[1], csr[2], rows)
        Ms = np.ascontiguousarray(prevA[rows])
        bs = orc.sum_by_cols(B1)
        if method == "pg":
            stepA = step_last * 0.5
            cnst_div = 1. / (1. + 2. * l2 * step_last)
            cs = bs * np.asarray(-stepA, bs.dtype) * np.asarray(-stepA, bs.dtype)
            with np.errstate(all="ignore"):
                orc.pg_iteration(Ms, B1, sd, sp, si, cnst_div, cs, None, stepA, 1.0, maxupd)
        elif method == "cg":
            orc.cg_iteration(Ms, B1, sd, sp, si, True, bs, l2, 1.0, maxupd)
        else:
            orc.tncg_iteration(Ms, B1, True, sd, sp, si, bs, l2, 1.0, maxupd, False)
        G = A1[rows]
        fin = np.isfinite(Ms) & np.isfinite(G)
        out["sample_rows"] = int(len(rows))
        out["sample_same_finite_pattern"] = bool(np.array_equal(np.isfinite(Ms), np.isfinite(G)))
        out["sample_max_scaled_err"] = float(np.max(np.abs(G[fin] - Ms[fin])) / max(float(np.max(np.abs(Ms[fin]))), 1e-300)) if fin.any() else None
        if method != "pg":
            from tests import helpers as H
            fo = H.half_objective(G, B1, sd, si, sp, bs, l2 if method == "cg" else 0.0)
            fr = H.half_objective(Ms, B1, sd, si, sp, bs, l2 if method == "cg" else 0.0)
            out["sample_objective_rel_diff"] = abs(fo - fr) / abs(fr)
    s.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
