"""Soak test of the multi-CU team kernels: N CG fp64 sweeps of the 1e8-nnz matrix, twice from the same start; the two runs
must agree bit for bit and finish without the team error word being set.   usage: soak_team.py [sweeps=20] [scale=1]
Two of them started together on one GPU (scale 4) exercise team launches of different processes competing for the CUs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 1 / scale of the matrix (same row lengths): for several processes at once
dimA, dimB = 10 ** 6 // scale, 10 ** 5 // scale
trip = synth.uniform_triplets(dimA, dimB, 10 ** 8 // scale, seed=1)
A0, B0 = harness.initialize_matrices(dimA, dimB, 50, False, 1)
res = []
for run in range(2):
    s = api.Session.from_coo(trip, 50, False)
    s.set_factors(A0, B0)
    p = s.make_params("cg", 1e4, maxupd=5)
    t0 = time.time()
    for _ in range(n):
        s.half_sweep(0, p, 1e-7, 1.0)
        s.half_sweep(1, p, 1e-7, 1.0)
    A, B = s.get_factors()       # raises if a team launch gave up
    print(f"run {run}: {n} sweeps in {time.time() - t0:.2f} s, finite {np.isfinite(A).all() and np.isfinite(B).all()}", flush=True)
    res.append((A, B))
    s.close()
print("bit-identical:", np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), "checksum", float(np.abs(res[0][0]).sum()))
