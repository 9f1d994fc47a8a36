"""Development: config C5 (tncg fp64, k = 100) in steady state: per-launch times of both halves and the evaluations the solvers made.
usage: c5_halves.py [warm sweeps = 2] [timed sweeps = 3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness, synth
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 2
timed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = synth.lastfm_like_coo()
dimA, dimB = c.shape
s = api.Session.from_coo(c, 100, False)
A0, B0 = harness.initialize_matrices(dimA, dimB, 100, False, 1)
s.set_factors(A0, B0)
p = s.make_params("tncg", 1e3, maxupd=1500, reuse_prev=True, early_stop=False)
step = 1e-7
for _ in range(warm):
    step = s.sweep(p, step)
s.profile(True)
for _ in range(timed):
    step = s.sweep(p, step)
for w in (0, 1):
    ms, n = s.kernel_time(w)
    d = s.decision_stats(w)
    print("half", "AB"[w ^ 1] if False else ("B" if w == 0 else "A"), "ms per half-sweep %.1f" % (ms / timed), "evaluations (last sweep) %d, nnz x evaluations %.3e" % (d["evaluations"], d["nnz_evaluations"]))
    for L in s.launch_profile(w):
        print("    %-70s rows=%-7d nnz=%-9d ms=%.1f" % (L["kernel"][:70], L["rows"], L["nnz"], L["ms"] / L["calls"]))
import hashlib
A, B = s.get_factors()
print("factors sha256", hashlib.sha256(A.tobytes()).hexdigest()[:16], hashlib.sha256(B.tobytes()).hexdigest()[:16], "(bit-identical builds print the same)")
