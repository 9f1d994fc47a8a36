"""Development probe (needs a -DPMF_PROBE build: POISMF_HIP_EXTRA_FLAGS=-DPMF_PROBE both when building and when running):
shader-clock stamps of the phases of one evaluation of the register engine, workgroup 5 of the first launch of a half
of the 1e8-nnz matrix.   usage: probe_phases.py [which=0|1] [method] [float=1|0]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness, synth
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
method = sys.argv[2] if len(sys.argv) > 2 else "pg"
use_float = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
trip = synth.uniform_triplets(10 ** 6, 10 ** 5, 10 ** 8, seed=1)
s = api.Session.from_coo(trip, 50, use_float)
A0, B0 = harness.initialize_matrices(10 ** 6, 10 ** 5, 50, use_float, 1)
s.set_factors(A0, B0)
p = s.make_params(method, 1e3, maxupd=10) if method == "pg" else s.make_params(method, 1e3)
s.profile(True)
s.half_sweep(which, p, 1e-9, 1.0)
print("kernel ms", s.kernel_time(which))
out = np.zeros(16 * 60, np.uint32)
s.lib.poismf_hip_debug_eval_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
assert s.lib.poismf_hip_debug_eval_rows(s.h, which, out.ctypes.data_as(C.c_void_p), len(out)) == 0
t = out.reshape(60, 16).astype(np.int64)
names = ["dots(b0)", "butterfly(b0)", "coef(b0)", "axpy(b0)", "other batches", "groups", "lds+barrier", "wave sums", "update"]
print("nnz | " + " | ".join(names) + " || eval total | row total")
rows = []
for i in range(2, 58):
    r = t[i]
    d = [int((r[j + 1] - r[j]) & 0xffffffff) for j in range(9)]
    rows.append(d + [sum(d[:8]), int((t[i + 1][10] - r[10]) & 0xffffffff)])
    if i < 12:
        print(int(r[11]), d, sum(d[:8]), rows[-1][-1])
print("median", [int(v) for v in np.median(np.array(rows), axis=0)])
