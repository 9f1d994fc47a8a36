#!/usr/bin/env python3
"""Time the device COO -> CSR + CSC conversion against SciPy on a BASELINE shape (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poismf_amd import api, harness, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
coo = synth.CONFIGS[cfg][0]()
t0 = time.time(); csr_h, csc_h = harness.process_data(coo, True); t_host = time.time() - t0
api.coo_to_csr_csc(synth.readme_coo(), True)  # warm the runtime
t0 = time.time(); csr_g, csc_g = api.coo_to_csr_csc(coo, True); t_gpu = time.time() - t0
same = all(np.array_equal(a, b) for a, b in zip(csr_g + csc_g, csr_h + csc_h))
print(f"{cfg}: {coo.nnz} triplets -> {len(csr_g[0])} nnz; SciPy {t_host:.2f} s, device (host buffers in and out) {t_gpu:.2f} s, identical={same}")
