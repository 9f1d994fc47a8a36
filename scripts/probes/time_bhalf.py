"""Development: per-launch times of the B half (PG fp32, finite hyper-parameters) of the 1e8-nnz matrix for a few maxupd values.
For variant builds that compile the lane kernels alone (-DPMF_LANE_ONLY: only the rows the lane engine takes are launched at all).  usage: time_bhalf.py [maxupd ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness, synth
ups = [int(v) for v in sys.argv[1:]] or [10]
trip = synth.uniform_triplets(10 ** 6, 10 ** 5, 10 ** 8, seed=1)
s = api.Session.from_coo(trip, 50, True)
A0, B0 = harness.initialize_matrices(10 ** 6, 10 ** 5, 50, True, 1)
for mu in ups:
    s.set_factors(A0, B0)
    p = s.make_params("pg", 1e3, maxupd=mu)
    for _ in range(2):
        s.half_sweep(0, p, 1e-9, 1.0)
    s.profile(True)
    for _ in range(5):
        s.half_sweep(0, p, 1e-9, 1.0)
    print("maxupd", mu, " ".join(f"{L['kernel'].split('<')[1][:40]} rows={L['rows']} ms={L['ms'] / L['calls']:.3f};" for L in s.launch_profile(0)))
    s.profile(False)
