"""Development: one half-sweep of one solver on a seeded matrix; saves the updated factor to /tmp/half_bits_<tag>.npy, or (second argument = the tag of
an earlier run) compares with it and prints which rows differ, by row length.  usage: half_bits.py <tag> [compare-with-tag] ; env HB_METHOD, HB_MAXUPD, HB_WHICH"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api, harness, synth
method = os.environ.get("HB_METHOD", "cg"); maxupd = int(os.environ.get("HB_MAXUPD", "1")); which = int(os.environ.get("HB_WHICH", "1"))
c = synth.uniform_triplets(100000, 10000, 10 ** 7, seed=1)
s = api.Session.from_coo(c, 50, True)
A0, B0 = harness.initialize_matrices(100000, 10000, 50, True, 1)
s.set_factors(A0, B0)
l2, _, _ = harness.auto_defaults(method, 50)
p = s.make_params(method, l2, maxupd=maxupd, limit_step=True, reuse_prev=False, early_stop=False)
s.half_sweep(which, p, 1e-7, 1.0)
print("PLAN", " ".join(name for name, _ in s.plan(which)))
A, B = s.get_factors()
M = A if which == 1 else B
np.save("/tmp/half_bits_%s.npy" % sys.argv[1], M)
if len(sys.argv) > 2:
    R = np.load("/tmp/half_bits_%s.npy" % sys.argv[2])
    import scipy.sparse as sp
    m = sp.coo_matrix((c.data, (c.row, c.col)), shape=c.shape)
    m.sum_duplicates()
    nnz = np.diff((m.tocsr() if which == 1 else m.tocsc()).indptr)
    bad = np.nonzero((M != R).any(axis=1))[0]
    print("rows that differ: %d of %d" % (len(bad), M.shape[0]))
    if len(bad):
        print("  lengths of differing rows: min %d max %d; first rows %s" % (nnz[bad].min(), nnz[bad].max(), [(int(r), int(nnz[r])) for r in bad[:6]]))
        r = bad[0]
        d = np.nonzero(M[r] != R[r])[0]
        print("  row %d: %d elements differ, e.g. %s" % (r, len(d), [(int(i), float(M[r, i]), float(R[r, i])) for i in d[:4]]))
