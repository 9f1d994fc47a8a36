#!/usr/bin/env python3
"""Development probe: CG fp32 on ~1000-nonzero rows, objective after 1..5 iterations: GPU vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
from oracle import bindings
from poismf_amd import api, harness, synth
from tests.test_gpu_fullsize import _row_objectives

dimA, dimB, n, k = 200000, 200, 200000, 50
trip = synth.uniform_triplets(dimA, dimB, n, seed=3)
use_float = True
csr, csc = harness.process_data(sp.coo_matrix((trip.data, (trip.row, trip.col)), shape=trip.shape), use_float)
A0, B0 = harness.initialize_matrices(dimA, dimB, k, use_float, 1)
l2 = 1e4
orc = bindings.Oracle(use_float)
bs = A0.astype(np.float64).sum(0).astype(A0.dtype)
s = api.Session(csr, csc, dimA, dimB, k, use_float)
for mu in (1, 2, 3, 4, 5):
    Bo = B0.copy()
    orc.cg_iteration(Bo, A0, csc[0], csc[2], csc[1], True, bs, l2, 1.0, mu)
    s.set_factors(A0, B0)
    s.half_sweep(0, s.make_params("cg", l2, maxupd=mu), 1e-7, 1.0)
    _, B1 = s.get_factors()
    fo = _row_objectives(Bo, A0, csc[0], csc[1], csc[2], bs, l2)
    fg = _row_objectives(B1, A0, csc[0], csc[1], csc[2], bs, l2)
    print(mu, "oracle", fo[:5], "\n   gpu   ", fg[:5], "\n   max |B1-Bo| / max|Bo| rows 0..4:", [float(np.abs(B1[r] - Bo[r]).max() / np.abs(Bo[r]).max()) for r in range(5)])
    if mu == 1:
        print("   B after 1 iteration, row 0: oracle", Bo[0][:6], "gpu", B1[0][:6])
