#!/usr/bin/env python3
"""Development aid: PG / CG half-sweep kernel times on a uniform dimA x dimB matrix with the given nnz (fp32, k = 50).
    python scripts/time_shape.py 200000 100000 20000000 [method] [maxupd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poismf_amd import api, harness, synth
dimA, dimB, nnz = (int(float(x)) for x in sys.argv[1:4])
method = sys.argv[4] if len(sys.argv) > 4 else "pg"
coo = synth.uniform_coo(dimA, dimB, nnz, seed=1)
csr, csc = harness.process_data(coo, True)
A0, B0 = harness.initialize_matrices(dimA, dimB, 50, True, 1)
s = api.Session(csr, csc, dimA, dimB, 50, True)
s.set_factors(A0, B0)
l2, mu, _ = harness.auto_defaults(method, 50)
if len(sys.argv) > 5: mu = int(sys.argv[5])
p = s.make_params(method, l2, maxupd=mu)
step = s.sweep(p, 1e-7)
s.profile(True)
n = 3
for _ in range(n): step = s.sweep(p, step)
kb, ka = s.kernel_time(0)[0] / n, s.kernel_time(1)[0] / n
nz = len(csr[0])
print(f"{dimA}x{dimB} nnz={nz} {method} maxupd={mu}: B half {kb:.3f} ms ({nz / dimB:.0f} nnz/row, {kb * 1e6 / nz:.3f} ns/nnz)  A half {ka:.3f} ms ({nz / dimA:.0f} nnz/row, {ka * 1e6 / nz:.3f} ns/nnz)")
