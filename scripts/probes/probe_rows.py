#!/usr/bin/env python3
"""Development probe: per-case numbers behind tests/test_gpu_rows.py (GPU vs golden vs oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import bindings
from tests import test_gpu_rows as T

for use_float in (False, True):
    z = np.load(os.path.join(T.GOLD, f"rows_{'f32' if use_float else 'f64'}.npz"))
    orc = bindings.Oracle(use_float)
    for ci in range(int(z["ncases"])):
        p, F, a, bsum, xval, xind, w = T._case(z, ci)
        l2 = float(z[p + "l2tn"])
        for reuse in (True, False):
            for mf in (10, 75, 750):
                x = T._one_row(F, bsum, a, xval, xind, w, l2_reg=l2, step_size=1e-7, niter=1, maxupd=mf, method="tncg", limit_step=False, reuse_mean=reuse)
                meta = z[p + f"tnc_{int(reuse)}_{mf}_meta"]
                a0 = a if reuse else np.full_like(a, 1e-3)
                xo, fo, nfo, nio, rco = orc.tnc_row(a0, F, bsum, xval, xind, l2, w, mf)
                fg = T._objective(x, F, bsum, xval, xind, 0.0, w)
                fr = T._objective(z[p + f"tnc_{int(reuse)}_{mf}_x"], F, bsum, xval, xind, 0.0, w)
                print(f"{'f32' if use_float else 'f64'} case {ci} k={F.shape[1]} nnz={len(xval)} w={w} reuse={reuse} maxnfeval={mf}: f gpu {fg:.8g} golden {float(meta[0]):.8g} (recomputed {fr:.8g}, nfeval {int(meta[1])} rc {int(meta[3])}) oracle {fo:.8g} (nfeval {nfo} rc {rco})")
