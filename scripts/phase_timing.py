#!/usr/bin/env python3
"""Development aid: per-phase shader-clock breakdown of half_sweep_kernel on config C2 (needs the -DPMF_TIMING
build: hipcc -DUSE_FLOAT -DPMF_TIMING ... -o scripts/libpoismf_hip_f_timing.so)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poismf_amd import api, build, harness, synth
here = os.path.dirname(os.path.abspath(__file__))
if os.path.exists(os.path.join(here, "libpoismf_hip_f_timing.so")):
    build.lib_path = lambda use_float: os.path.join(here, "libpoismf_hip_f_timing.so")
# otherwise: the in-tree library, built with POISMF_HIP_EXTRA_FLAGS=-DPMF_TIMING python -m poismf_amd.build --force
maxupd = int(sys.argv[1]) if len(sys.argv) > 1 else 10
method = sys.argv[2] if len(sys.argv) > 2 else "pg"
UF = not os.environ.get("PMF_FP64")   # PMF_FP64=1: the double-precision library
coo = synth.uniform_coo(10 ** 5, 10 ** 5, 10 ** 7, seed=1)
csr, csc = harness.process_data(coo, UF)
A0, B0 = harness.initialize_matrices(10 ** 5, 10 ** 5, 50, UF, 1)
s = api.Session(csr, csc, 10 ** 5, 10 ** 5, 50, UF)
s.set_factors(A0, B0)
l2, mu, _ = harness.auto_defaults(method, 50)
p = s.make_params(method, l2, maxupd=maxupd)
lib = s.lib
lib.poismf_hip_debug_timing.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 8)()
step = s.sweep(p, 1e-7)
lib.poismf_hip_debug_timing(buf)
s.profile(True)
n = 3
for _ in range(n):
    step = s.sweep(p, step)
lib.poismf_hip_debug_timing(buf)
t = np.array(list(buf)[:6], dtype=np.float64)
names = ["gather", "phase1", "coef/div", "phase2", "combine", "kernel(total wave-cycles)"]
if os.environ.get("PMF_ROW_TIMERS"):   # register engine: row-level timers of sweep_rows
    names = ["row hand-out", "gather (tile landed)", "solver", "store", "-", "kernel(total wave-cycles)"]
km = sum(s.kernel_time(w)[0] for w in (0, 1)) / n
print(f"method={method} maxupd={maxupd}: kernel {km:.3f} ms per sweep")
for nm, v in zip(names, t):
    print(f"  {nm:28s} {v / n:14.4g} wave-cycles per sweep  ({100 * v / t[5]:5.1f} % of wave time)")
print(f"  {'other':28s} {(t[5] - t[:5].sum()) / n:14.4g}                       ({100 * (t[5] - t[:5].sum()) / t[5]:5.1f} %)")
