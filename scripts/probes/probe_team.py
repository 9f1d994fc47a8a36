"""Development probe (needs a -DPMF_PROBE build: POISMF_HIP_EXTRA_FLAGS=-DPMF_PROBE both when building and when running):
per item row of the 1e8-nnz matrix under CG fp64 (team kernels), shader cycles of the whole row and of the time its
first member's wave 0 spent waiting for the other members' granules."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness, synth
trip = synth.uniform_triplets(10 ** 6, 10 ** 5, 10 ** 8, seed=1)
s = api.Session.from_coo(trip, 50, False)
A0, B0 = harness.initialize_matrices(10 ** 6, 10 ** 5, 50, False, 1)
s.set_factors(A0, B0)
p = s.make_params("cg", 1e4, maxupd=5)
for _ in range(2):        # steady state: the first sweeps from the random start take fewer evaluations per row
    s.half_sweep(0, p, 1e-7, 1.0)
    s.half_sweep(1, p, 1e-7, 1.0)
s.profile(True)
s.half_sweep(0, p, 1e-7, 1.0)
print("kernel ms", s.kernel_time(0), s.plan(0))
out = np.zeros(10 ** 5, np.uint32)
s.lib.poismf_hip_debug_eval_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
assert s.lib.poismf_hip_debug_eval_rows(s.h, 0, out.ctypes.data_as(C.c_void_p), len(out)) == 0
names = ["evaluations (incl. combine)", "combine_waves (LDS + team)", "line-search batches", "unpark", "gather until landed", "rows total", "team wait", "combine_scalars (inside the line-search batches)"]
head = np.zeros(16, np.uint64)
s.lib.poismf_hip_debug_team_head.argtypes = [C.c_void_p, C.c_void_p]
assert s.lib.poismf_hip_debug_team_head(s.h, head.ctypes.data_as(C.c_void_p)) == 0
acc = head[8:16].astype(np.float64)
print("cycles of the leaders' wave 0 summed over the LAST team launch, share of rows total:")
for n, v in zip(names, acc):
    print(f"  {n:32s} {v:14.0f}  {v / acc[5]:.3f}")
wait = (out & 0xffff).astype(np.float64) * 256
total = (out >> 16).astype(np.float64) * 256
ok = total > 0
print("rows stamped", int(ok.sum()), "mean cycles per row", total[ok].mean(), "of which waiting", wait[ok].mean(),
      "quartiles total", np.percentile(total[ok], [5, 25, 50, 75, 95]), "wait", np.percentile(wait[ok], [5, 25, 50, 75, 95]))
