#!/usr/bin/env python3
"""Development aid: wall time of the drop-in run_poismf() (host buffers in, host buffers out) on config C2 and on the
1e8-nnz matrix: set-up (upload, index narrowing and row sort on the device) vs sweeps.  POISMF_HIP_VERBOSE=1 makes the
library print its own phase times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("POISMF_HIP_VERBOSE", "1")
import numpy as np
from poismf_amd import api, harness, synth
for name, (dimA, dimB, n) in (("C2", (10 ** 5, 10 ** 5, 10 ** 7)), ("C4", (10 ** 6, 10 ** 5, 10 ** 8))):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    trip = synth.uniform_triplets(dimA, dimB, n, seed=1)
    csr, csc = api.coo_to_csr_csc(trip, True)
    del trip
    A0, B0 = harness.initialize_matrices(dimA, dimB, 50, True, 1)
    res = {}
    for numiter in (1, 1, 11):
        A, B = A0.copy(), B0.copy()
        t = time.perf_counter()
        api._run_poismf(csr[0], csr[1], csr[2], csc[0], csc[1], csc[2], A, B, "pg", True, 1e9, 0., 1., 1e-7, numiter, 1, False, True, True, 1)
        res[numiter] = time.perf_counter() - t
        print(f"{name}: run_poismf numiter={numiter}: {res[numiter] * 1e3:.1f} ms", flush=True)
    print(f"{name}: abi_ms_first_iter {res[1] * 1e3:.1f}  abi_ms_per_extra_iter {(res[11] - res[1]) / 10 * 1e3:.2f}", flush=True)
