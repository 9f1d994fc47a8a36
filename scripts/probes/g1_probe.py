import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers as H
from tests.test_gpu_rows import _row_eval
for use_float in (False,):
    orc = H.checker(use_float, "cg")
    for nnz in (1000, 1024, 1025, 1030, 1040, 1088, 1100):
        F, a, bsum, xval, xind = H.random_row(50, nnz, 4000, use_float, seed=7 + nnz)
        xind = np.ascontiguousarray(xind, dtype=np.uint64)
        for w in (1.0,):
            f0, g0 = _row_eval(F, bsum, a, xval, xind, w, 1e4, 0)
            f1, g1 = _row_eval(F, bsum, a, xval, xind, w, 1e3, 1)
            rf0 = orc.calc_fun_single(a, F, bsum, xval, xind, 1e4, w)
            rg0 = orc.calc_grad_single(a, F, bsum, xval, xind, 1e4, w, w != 1.0)
            rf1, rg1 = orc.calc_fun_and_grad(a, F, bsum, xval, xind, 1e3, w)
            print(nnz, "f0", f0, rf0, "f1", f1, rf1, "g0 err", np.max(np.abs(g0 - rg0)) / np.max(np.abs(rg0)), "g1 err", np.max(np.abs(g1 - rg1)) / np.max(np.abs(rg1)))
