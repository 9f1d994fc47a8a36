"""Rows of 1025 .. 1088 nonzeros (the partial-LDS-set lane instance): is a run bit-identical to its repeat, per solver?"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_regtile import ragged_problem
from tests import helpers as H
from poismf_amd import api, harness
k = 50
lengths = [1024, 1025, 1026, 1030, 1040, 1041, 1056, 1072, 1087, 1088] * 30
csr, csc, A0, B0 = ragged_problem(lengths, 4000, k, False, seed=5)
val, ind, ptr = csr
for method, maxupd in (("cg", 5), ("tncg", 40), ("pg", 1)):
    l2, _, _ = harness.auto_defaults(method, k)
    orc = H.checker(False, method)
    bs = orc.sum_by_cols(B0)
    outs = []
    for rep in range(3):
        A, ni, nf, rc = api.factors_multiple_with_decisions(B0, bs, A0.mean(axis=0), ptr, ind, val, l2_reg=l2, niter=1, maxupd=maxupd, method=method,
                                                            limit_step=1, reuse_mean=0)
        outs.append((A.copy(), nf.copy()))
    lens = np.diff(ptr.astype(np.int64))[:len(lengths)]
    for rep in (1, 2):
        diff = np.flatnonzero((outs[rep][0][:len(lengths)] != outs[0][0][:len(lengths)]).any(axis=1))
        print(method, "repeat", rep, "rows whose bits differ:", len(diff), sorted(set(lens[diff].tolist())))
