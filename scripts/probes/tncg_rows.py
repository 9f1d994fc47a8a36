"""Per-row look at fp64 TNCG on the hand-over lengths: one iteration, rows of A against the oracle."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_regtile import ragged_problem, BOUNDARY_LENGTHS
from tests.test_gpu_parity import gpu_run, oracle_run
k = int(sys.argv[1]) if len(sys.argv) > 1 else 50
csr, csc, A0, B0 = ragged_problem(BOUNDARY_LENGTHS, 4000, k, False, seed=11)
for method, kw in (("tncg", dict(maxupd=300)), ("cg", {})):
    A, B, args = gpu_run(csr, csc, A0, B0, method, 1, k, **kw)
    Ar, Br = oracle_run(False, csr, csc, A0, B0, method, args)
    for name, X, Xr, ptr in (("A", A, Ar, csr[2]), ("B", B, Br, csc[2])):
        d = np.abs(X - Xr).max(axis=1) / np.abs(Xr).max()
        bad = np.nonzero(d > 1e-4)[0]
        print(method, name, "rows off by > 1e-4:", [(int(r), int(ptr[r + 1] - ptr[r]), float(d[r])) for r in bad][:20])
    if method == "tncg":
        ind, ptr, val = csr[1], csr[2], csr[0]
        bsum = B0.sum(axis=0)
        def obj(a, r):
            j = ind[ptr[r]:ptr[r + 1]].astype(np.int64)
            return float(a @ bsum - val[ptr[r]:ptr[r + 1]] @ np.log(B0[j] @ a) + args["l2_reg"] * (a @ a))
        for r in range(40, len(ptr) - 2):
            print(r, int(ptr[r + 1] - ptr[r]), "f_gpu - f_oracle = %.3e  (f = %.6e)" % (obj(A[r], r) - obj(Ar[r], r), obj(Ar[r], r)))
