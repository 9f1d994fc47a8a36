"""Development: config C5's A half alone (355 k user rows of <= 64 nonzeros + 3 k of 65 .. 128, tncg fp64 k = 100: lane-engine launches only, so
-DPMF_LANE_ONLY variant builds can run it).  `save`: two warm sweeps with this build, factors written to /tmp/c5_warm.npz; otherwise: the
factors are read from there, three A halves are timed per launch and a hash of A is printed (bit-identical builds print the same).
usage: c5_ahalf.py [save]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness, synth
c = synth.lastfm_like_coo()
dimA, dimB = c.shape
s = api.Session.from_coo(c, 100, False)
p = s.make_params("tncg", 1e3, maxupd=1500, reuse_prev=True, early_stop=False)
if len(sys.argv) > 1 and sys.argv[1] == "save":
    A0, B0 = harness.initialize_matrices(dimA, dimB, 100, False, 1)
    s.set_factors(A0, B0)
    step = 1e-7
    for _ in range(2):
        step = s.sweep(p, step)
    A, B = s.get_factors()
    np.savez("/tmp/c5_warm.npz", A=A, B=B)
    sys.exit(0)
w = np.load("/tmp/c5_warm.npz")
s.set_factors(w["A"], w["B"])
s.profile(True)
n = 3
for _ in range(n):
    s.half_sweep(1, p, 1e-7, 1.0)
ms, _ = s.kernel_time(1)
d = s.decision_stats(1)
print("A half ms %.1f" % (ms / n), "evaluations (last half) %d" % d["evaluations"])
for L in s.launch_profile(1):
    print("    %-70s rows=%-7d nnz=%-9d ms=%.1f" % (L["kernel"][:70], L["rows"], L["nnz"], L["ms"] / L["calls"]))
A, B = s.get_factors()
print("A sha256", hashlib.sha256(A.tobytes()).hexdigest()[:16])
