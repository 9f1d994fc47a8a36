"""Development probe (needs a -DPMF_PROBE build): shader-clock stamps of the phases of one evaluation (the 4th of a row) of the
lane-per-nonzero engine, one workgroup of the first launch of the B half (PG fp32, finite hyper-parameters) of the 1e8-nnz matrix.
usage: probe_lane.py [maxupd=10]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness, synth
maxupd = int(sys.argv[1]) if len(sys.argv) > 1 else 10
trip = synth.uniform_triplets(10 ** 6, 10 ** 5, 10 ** 8, seed=1)
s = api.Session.from_coo(trip, 50, True)
A0, B0 = harness.initialize_matrices(10 ** 6, 10 ** 5, 50, True, 1)
s.set_factors(A0, B0)
p = s.make_params("pg", 1e3, maxupd=maxupd)
s.profile(True)
s.half_sweep(0, p, 1e-9, 1.0)
print("kernel ms", s.kernel_time(0), s.plan(0))
out = np.zeros(16 * 60, np.uint32)
s.lib.poismf_hip_debug_eval_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
assert s.lib.poismf_hip_debug_eval_rows(s.h, 0, out.ctypes.data_as(C.c_void_p), len(out)) == 0
t = out.reshape(60, 16).astype(np.int64)
names = ["dots", "coef", "reduce", "combine", "update(to stamp 9)", "| reduce: batch0", "batch1", "batch2", "batch3", "swaps"]
print("nnz | " + " | ".join(names) + " || row total (start to next row's start)")
rows = []
for i in range(2, 58):
    r = t[i]
    d = [int((r[j + 1] - r[j]) & 0xffffffff) for j in range(4)] + [int((r[9] - r[4]) & 0xffffffff)]
    d += [int((r[5] - r[2]) & 0xffffffff), int((r[6] - r[5]) & 0xffffffff), int((r[7] - r[6]) & 0xffffffff), int((r[8] - r[7]) & 0xffffffff), int((r[3] - r[8]) & 0xffffffff)]
    rows.append(d + [int((t[i + 1][10] - r[10]) & 0xffffffff)])
    if i < 10:
        print(int(r[11]), rows[-1])
print("median", [int(v) for v in np.median(np.array(rows), axis=0)])
