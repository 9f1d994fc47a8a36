"""fp64 TNCG decisions (nfeval, niter, rc) and final objective per row of the hand-over lengths: device against the checker."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_regtile import ragged_problem, BOUNDARY_LENGTHS
from tests import helpers as H
from poismf_amd import api, harness
k = int(sys.argv[1]) if len(sys.argv) > 1 else 50
maxupd = int(sys.argv[2]) if len(sys.argv) > 2 else 300
csr, csc, A0, B0 = ragged_problem(BOUNDARY_LENGTHS, 4000, k, False, seed=11)
l2, _, _ = harness.auto_defaults("tncg", k)
orc = H.checker(False, "tncg")
bs = orc.sum_by_cols(B0)
val, ind, ptr = csr
A, ni, nf, rc = api.factors_multiple_with_decisions(B0, bs, A0.mean(axis=0), ptr, ind, val, l2_reg=l2, niter=1, maxupd=maxupd, method="tncg",
                                                    limit_step=0, reuse_mean=0)
# reuse_mean=0: every row starts from Amean?  -- compare against the checker started from the same point
start = A0.mean(axis=0)
def obj(a, r):
    j = ind[ptr[r]:ptr[r + 1]].astype(np.int64)
    return float(a @ bs - val[int(ptr[r]):int(ptr[r + 1])] @ np.log(B0[j] @ a) + l2 * (a @ a))
for r in range(len(ptr) - 2):
    xv, xi = np.ascontiguousarray(val[int(ptr[r]):int(ptr[r + 1])]), np.ascontiguousarray(ind[int(ptr[r]):int(ptr[r + 1])])
    xr, f_r, nf_r, ni_r, rc_r = orc.tnc_row(start, B0, bs, xv, xi, l2, 1.0, maxupd)
    flag = "" if (int(nf[r]), int(ni[r]), int(rc[r])) == (nf_r, ni_r, rc_r) else "   <--"
    print(f"{r:3d} nnz {len(xv):5d} gpu (nf {int(nf[r]):3d} ni {int(ni[r]):3d} rc {int(rc[r])}) ref (nf {nf_r:3d} ni {ni_r:3d} rc {rc_r})  f_gpu - f_ref {obj(A[r], r) - obj(xr, r):+.3e}{flag}")
