#!/usr/bin/env python3
"""Development probe: CG fp32 on ~1000-nonzero rows (the 8-wave register kernel): GPU vs oracle fp32 vs oracle fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import bindings
from poismf_amd import api, harness, synth
from tests import helpers as H
from tests.test_gpu_fullsize import _row_objectives

dimA, dimB, n, k = 200000, 200, 200000, 50
if len(sys.argv) > 1:
    dimA = int(sys.argv[1])
trip = synth.uniform_triplets(dimA, dimB, n, seed=3)
res = {}
for use_float in (True, False):
    csr, csc = harness.process_data(__import__("scipy.sparse").sparse.coo_matrix((trip.data, (trip.row, trip.col)), shape=trip.shape), use_float)
    A0, B0 = harness.initialize_matrices(dimA, dimB, k, use_float, 1)
    l2, maxupd, _ = harness.auto_defaults("cg", k)
    orc = bindings.Oracle(use_float)
    bs = A0.astype(np.float64).sum(0).astype(A0.dtype)
    Bo = B0.copy()
    orc.cg_iteration(Bo, A0, csc[0], csc[2], csc[1], True, bs, l2, 1.0, maxupd)
    res[("oracle", use_float)] = _row_objectives(Bo, A0, csc[0], csc[1], csc[2], bs, l2)
    for knob in ("", "POISMF_HIP_NO_REGTILE"):
        if knob:
            os.environ[knob] = "1"
        s = api.Session(csr, csc, dimA, dimB, k, use_float)
        s.set_factors(A0, B0)
        p = s.make_params("cg", l2, maxupd=maxupd)
        s.half_sweep(0, p, 1e-7, 1.0)
        _, B1 = s.get_factors()
        print(s.plan(0))
        s.close()
        res[("gpu" + knob, use_float)] = _row_objectives(B1, A0, csc[0], csc[1], csc[2], bs, l2)
        if knob:
            del os.environ[knob]
f0 = _row_objectives(B0, A0, csc[0], csc[1], csc[2], bs, l2)
print("row lengths", np.diff(csc[2].astype(np.int64))[:8])
print("start     ", f0[:6])
for key, v in res.items():
    print(key, v[:6], "sum %.10g" % v.sum())
