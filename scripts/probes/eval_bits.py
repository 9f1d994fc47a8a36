"""Development: gradients of a few seeded fp32 k = 50 rows through the evaluation-only kernels (poismf_hip_debug_row_eval); prints a hash per row --
two builds whose evaluation is bit-identical print the same lines.  usage: eval_bits.py [k = 50] [fp32 = 1]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from poismf_amd import api
k = int(sys.argv[1]) if len(sys.argv) > 1 else 50
uf = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dt = np.float32 if uf else np.float64
rng = np.random.default_rng(5)
dimB = 20000
F = (rng.gamma(1.0, 0.3, (dimB, k)) + 0.01).astype(dt)
lengths = [3, 40, 64, 100, 128, 129, 200, 256, 300, 500, 1000, 1100]
indptr = np.zeros(len(lengths) + 1, dtype=np.uint64)
indptr[1:] = np.cumsum(lengths)
ind = np.concatenate([rng.choice(dimB, n, replace=False) for n in lengths]).astype(np.uint64)
val = (1.0 + np.floor(rng.gamma(1.0, 1.0, len(ind)))).astype(dt)
bsum = F.sum(0).astype(dt)
point = (rng.gamma(1.0, 0.3, (len(lengths), k)) + 0.01).astype(dt)
scale = float(os.environ.get("EB_SCALE", "1"))   # e.g. 1e-22: the tile's products in the denormal range of floats
F = (F * scale).astype(dt); bsum = F.sum(0).astype(dt)
for which in (0, 1):
    f, G = api.debug_row_eval(F, bsum, point, indptr, ind, val, 1e3, 1.0, which)
    for r, n in enumerate(lengths):
        print("which %d nnz %5d  f %.9g  g sha %s" % (which, n, f[r], hashlib.sha256(G[r].tobytes()).hexdigest()[:12]))
