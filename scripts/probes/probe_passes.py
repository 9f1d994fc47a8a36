import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from poismf_amd import api, harness, synth
uf = sys.argv[1] == "f32"
trip = synth.uniform_triplets(10 ** 6 // 4, 10 ** 5 // 4, 10 ** 8 // 4, seed=1)
s = api.Session.from_coo(trip, 50, uf)
A0, B0 = harness.initialize_matrices(10 ** 6 // 4, 10 ** 5 // 4, 50, uf, 1)
s.set_factors(A0, B0)
l2, mu, _ = harness.auto_defaults("cg", 50)
p = s.make_params("cg", l2, maxupd=mu, limit_step=True, reuse_prev=True)
for _ in range(2):
    s.half_sweep(0, p, 1e-7, 1.0); s.half_sweep(1, p, 1e-7, 1.0)
s.profile(True)
s.half_sweep(0, p, 1e-7, 1.0); s.half_sweep(1, p, 1e-7, 1.0)
for w in (0, 1):
    print("half", w, "kernel ms", s.kernel_time(w), "eval stats", s.eval_stats(w) if hasattr(s, "eval_stats") else None, s.plan(w)[:3])
