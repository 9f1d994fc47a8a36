import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_gpu_giant import LENGTHS
from tests.test_gpu_regtile import ragged_problem
from tests.test_gpu_parity import gpu_run
k = int(sys.argv[2]); mu = int(sys.argv[3])
csr, csc, A0, B0 = ragged_problem(LENGTHS, 12000, k, False, seed=33)
A, B, args = gpu_run(csr, csc, A0, B0, "tncg", 1, k, maxupd=mu)
np.save(sys.argv[1], A)
