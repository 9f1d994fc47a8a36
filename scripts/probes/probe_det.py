import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness
from tests import helpers as H
dimA, dimB, k = 1500, 900, 50
for prec in (True, False):
    csr, csc, A0, B0 = H.small_problem(dimA, dimB, 60000, k, prec, seed=8, powerlaw=True, empty_rows=(5,))
    for method in ("tncg", "cg"):
        outs = []
        for rep in range(3):
            s = api.Session(csr, csc, dimA, dimB, k, prec)
            s.set_factors(A0, B0)
            p = s.make_params(method, 1e3, maxupd=40)
            if rep == 2:
                s.set_segments(0, 3)
                for j in range(3):
                    s.half_sweep(0, p, 1e-7, 1.0, seg=j)
            else:
                s.half_sweep(0, p, 1e-7, 1.0)
            A, B = s.get_factors()
            outs.append(B)
            plan = s.plan(0)
            s.close()
        lens = np.diff(csc[2].astype(np.int64))
        d01 = np.flatnonzero((outs[0] != outs[1]).any(axis=1)); d02 = np.flatnonzero((outs[0] != outs[2]).any(axis=1))
        print("   max |run0-run1|", float(np.abs(outs[0].astype(np.float64) - outs[1]).max()), "max |run0-seg|", float(np.abs(outs[0].astype(np.float64) - outs[2]).max()), "max|B|", float(np.abs(outs[0]).max()), "finite", np.isfinite(outs[0]).all(), np.isfinite(outs[1]).all())
        print("f32" if prec else "f64", method, "rows differing run0 vs run1:", len(d01), " run0 vs segmented:", len(d02), "lengths of those rows:", lens[d02][:12], plan[:2])
