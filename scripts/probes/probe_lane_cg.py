"""Development probe (needs a -DPMF_PROBE build): shader-clock stamps of the phases of one gradient evaluation (the 4th evaluation of a
row) of the lane-per-nonzero engine under CG, one workgroup of the first launch of a half of the 1e8-nnz matrix.
usage: probe_lane_cg.py [which=1] [float=0]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness, synth
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
use_float = (sys.argv[2] != "0") if len(sys.argv) > 2 else False
trip = synth.uniform_triplets(10 ** 6, 10 ** 5, 10 ** 8, seed=1)
s = api.Session.from_coo(trip, 50, use_float)
A0, B0 = harness.initialize_matrices(10 ** 6, 10 ** 5, 50, use_float, 1)
s.set_factors(A0, B0)
p = s.make_params("cg", 1e4, maxupd=5)
s.profile(True)
s.half_sweep(which, p, 1e-7, 1.0)
print("kernel ms", s.kernel_time(which), s.plan(which))
out = np.zeros(16 * 60, np.uint32)
s.lib.poismf_hip_debug_eval_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
assert s.lib.poismf_hip_debug_eval_rows(s.h, which, out.ctypes.data_as(C.c_void_p), len(out)) == 0
t = out.reshape(60, 16).astype(np.int64)
print("nnz | dots | coef | reduce | combine || row total (start to next row's start)   [the 4th eval() call of the row]")
rows = []
for i in range(2, 58):
    r = t[i]
    d = [int((r[j + 1] - r[j]) & 0xffffffff) for j in range(4)]
    rows.append(d + [int((t[i + 1][10] - r[10]) & 0xffffffff)])
    if i < 10:
        print(int(r[11]), rows[-1])
print("median", [int(v) for v in np.median(np.array(rows), axis=0)])
