"""Development: a handful of k = 100 fp64 rows of lane-team lengths, one TNCG half-sweep with two evaluations, against the same under
POISMF_HIP_NO_LANE_TEAMS=1 (a child process).  usage: lane_team_dbg.py [maxupd = 2]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
from poismf_amd import api, harness

MODE = os.environ.get("LT_MODE", "half1")
def problem(lengths, dimB, k, seed):
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for r, n in enumerate(lengths):
        rows.append(np.full(n, r)); cols.append(rng.choice(dimB, size=n, replace=False))
    row, col = np.concatenate(rows), np.concatenate(cols)
    val = 1.0 + np.floor(rng.gamma(1.0, 1.0, len(row)))
    coo = sp.coo_matrix((val, (row, col)), shape=(len(lengths) + 1, dimB))
    if MODE.startswith("T"): coo = coo.T.tocoo()
    csr, csc = harness.process_data(coo, False)
    A0, B0 = harness.initialize_matrices(coo.shape[0], coo.shape[1], k, False, seed + 1)
    return csr, csc, A0, B0

MODE = os.environ.get("LT_MODE", "half1")   # half1 | Thalf0 | sweep | Tsweep
maxupd = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lengths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [40, 100, 300, 700, 1000, 2500, 5000] + [1500] * 6
if os.environ.get("LT_POISON"):   # fill the device memory the library is about to get with a pattern (0xFF: NaNs, tag 0xffffffff)
    import torch
    t = [torch.full((1 << 30,), int(os.environ["LT_POISON"]), dtype=torch.uint8, device="cuda") for _ in range(24)]
    torch.cuda.synchronize(); del t; torch.cuda.empty_cache(); torch.cuda.synchronize()
csr, csc, A0, B0 = problem(lengths, 12000, 100, 33)
s = api.Session(csr, csc, A0.shape[0], B0.shape[0], 100, False)
p = s.make_params("tncg", 1e3, maxupd=maxupd)
if MODE == "half1" and not os.environ.get("LT_CHILD"):
    outs = []
    for _ in range(4):
        s.set_factors(A0, B0)
        s.half_sweep(1, p, 1e-7, 1.0)
        outs.append(s.get_factors()[0].copy())
    print("repeat diffs", [float(np.abs(o - outs[0]).max()) for o in outs[1:]])
s.set_factors(A0, B0)
if MODE == "half1": s.half_sweep(1, p, 1e-7, 1.0)
elif MODE == "Thalf0": s.half_sweep(0, p, 1e-7, 1.0)
else:
    step = 1e-7
    for _ in range(2): step = s.sweep(p, step)
print("PLAN", " ".join(name for name, _ in s.plan(0 if MODE.startswith("T") else 1)))
A, B = s.get_factors()
if MODE.startswith("T"): A = B
s.close()
print("start row 0:", " ".join("%.15g" % v for v in (B0 if MODE.startswith("T") else A0)[0, :4]))
print("LT_CHILD" if os.environ.get("LT_CHILD") else "TEAM", "row 0:", " ".join("%.15g" % v for v in A[0, :4]), "| row", len(lengths) - 1, ":", " ".join("%.15g" % v for v in A[len(lengths) - 1, :3]))
if os.environ.get("LT_CHILD"):
    np.save(os.environ["LT_CHILD"], A)
else:
    e = dict(os.environ); e["POISMF_HIP_NO_LANE_TEAMS"] = "1"; e["LT_CHILD"] = "/tmp/lt_child.npy"
    subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], check=True, env=e)
    R = np.load("/tmp/lt_child.npy")
    for r, n in enumerate(lengths):
        d = np.abs(A[r] - R[r]).max() / max(1e-300, np.abs(R[r]).max())
        print("row %2d nnz %5d  max rel diff %.3g %s" % (r, n, d, "" if d < 1e-10 else "<<<<"))
