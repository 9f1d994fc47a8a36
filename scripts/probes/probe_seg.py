import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness
from tests import helpers as H
dimA, dimB, k = 1500, 900, 50
prec = False
csr, csc, A0, B0 = H.small_problem(dimA, dimB, 60000, k, prec, seed=8, powerlaw=True, empty_rows=(5,))
l2, maxupd, _ = harness.auto_defaults("pg", k)
for nseg in (1, 3):
    s = api.Session(csr, csc, dimA, dimB, k, prec)
    s.set_factors(A0, B0)
    p = s.make_params("pg", l2, maxupd=maxupd)
    if nseg > 1:
        print("segments", s.set_segments(0, nseg), [s.segment_rows(0, j) for j in range(nseg)])
        for j in range(nseg):
            s.half_sweep(0, p, 1e-7, 1.0, seg=j)
            A, B = s.get_factors()
            print(" after seg", j, "B finite", np.isfinite(B).all(), "B changed rows", int((B != B0).any(axis=1).sum()), "max", np.nanmax(B), s.plan(0))
    else:
        s.half_sweep(0, p, 1e-7, 1.0)
        A, B = s.get_factors()
        print("unsegmented: B finite", np.isfinite(B).all(), "B changed rows", int((B != B0).any(axis=1).sum()), "max", np.nanmax(B), s.plan(0))
    s.close()
