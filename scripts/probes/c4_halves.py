"""Development: one solver on the metric's matrix (uniform 1M x 100K, 1e8 triplets, k = 50) in steady state: per-launch times of both halves,
the evaluations the solvers made and a hash of the factors (bit-identical builds print the same).
usage: c4_halves.py [method = tncg] [fp32 = 1] [warm sweeps = 2] [timed sweeps = 3]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poismf_amd import api, harness, synth
method = sys.argv[1] if len(sys.argv) > 1 else "tncg"
uf = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
timed = int(sys.argv[4]) if len(sys.argv) > 4 else 3
f = int(os.environ.get("POISMF_BENCH_SCALE", "1"))
dimA, dimB, ntrip = 10 ** 6 // f, 10 ** 5 // f, 10 ** 8 // f
c = synth.uniform_triplets(dimA, dimB, ntrip, seed=1)
s = api.Session.from_coo(c, 50, uf)
A0, B0 = harness.initialize_matrices(dimA, dimB, 50, uf, 1)
s.set_factors(A0, B0)
l2, maxupd, _ = harness.auto_defaults(method, 50)
p = s.make_params(method, l2, maxupd=maxupd, limit_step=True, reuse_prev=True, early_stop=False)
step = 1e-7
for _ in range(warm):
    step = s.sweep(p, step)
s.profile(True)
for _ in range(timed):
    step = s.sweep(p, step)
tot = 0.0
for w in (0, 1):
    ms, n = s.kernel_time(w)
    tot += ms / timed
    d = s.decision_stats(w)
    print("half", "B" if w == 0 else "A", "ms per half-sweep %.2f" % (ms / timed), "evaluations (last sweep) %d, nnz x evaluations %.3e" % (d["evaluations"], d["nnz_evaluations"]))
    for L in s.launch_profile(w):
        print("    %-70s rows=%-7d nnz=%-9d ms=%.2f" % (L["kernel"][:70], L["rows"], L["nnz"], L["ms"] / L["calls"]))
print("kernel ms per sweep %.2f" % tot)
A, B = s.get_factors()
print("factors sha256", hashlib.sha256(A.tobytes()).hexdigest()[:16], hashlib.sha256(B.tobytes()).hexdigest()[:16])
