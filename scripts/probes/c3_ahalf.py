"""Development: config C3's A half alone (CG fp64, k = 50, 1M user rows of ~100 nonzeros: lane-engine launches only, so -DPMF_LANE_ONLY variant
builds can run it): three A halves from the starting factors, per-launch times and a hash of A.   usage: c3_ahalf.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poismf_amd import api, harness, synth
c = synth.uniform_triplets(10 ** 6, 10 ** 5, 10 ** 8, seed=1)
s = api.Session.from_coo(c, 50, False)
A0, B0 = harness.initialize_matrices(10 ** 6, 10 ** 5, 50, False, 1)
s.set_factors(A0, B0)
l2, maxupd, _ = harness.auto_defaults("cg", 50)
p = s.make_params("cg", l2, maxupd=maxupd, limit_step=True)
s.half_sweep(1, p, 1e-7, 1.0)
s.profile(True)
n = 3
for _ in range(n):
    s.half_sweep(1, p, 1e-7, 1.0)
ms, _ = s.kernel_time(1)
print("A half ms %.2f" % (ms / n))
for L in s.launch_profile(1):
    print("    %-70s rows=%-7d nnz=%-9d ms=%.2f" % (L["kernel"][:70], L["rows"], L["nnz"], L["ms"] / L["calls"]))
A, B = s.get_factors()
print("A sha256", hashlib.sha256(A.tobytes()).hexdigest()[:16])
