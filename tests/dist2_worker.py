"""Worker of tests/test_gpu_dist2.py: one rank of a 2-process, one-GPU run of the sharded driver (gloo between the
processes, both sessions on cuda:0).  Rank 0 writes the final factors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from poismf_amd import dist as pdist, harness  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    out, method, use_float = sys.argv[1], sys.argv[2], sys.argv[3] == "f32"
    early = len(sys.argv) > 4 and sys.argv[4] == "early"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dimA, dimB, k = 3000, 2000, 50
    csr, csc, A0, B0 = H.small_problem(dimA, dimB, 120000, k, use_float, seed=5, powerlaw=True, empty_rows=(3, 2999))
    rangesA = pdist.balanced_ranges(csr[2], world)      # unequal ranges: the per-owner broadcast path
    rangesB = pdist.equal_ranges(dimB, world)
    l2, maxupd, _ = harness.auto_defaults(method, k)
    be = pdist.HipBackend(csr, csc, dimA, dimB, k, use_float,
                          dict(method=method, l2_reg=l2, maxupd=60 if method == "tncg" else maxupd, limit_step=True, early_stop=early,
                               reuse_prev=True),
                          rangesA[rank], rangesB[rank], 0, segments=(2, 3))   # B half in 2, A half in 3 segments, exchanged one by one
    be.sess.set_factors(A0, B0)
    alt = pdist.ShardedAlternation(be, rangesA, rangesB, method, l2, 1e-7, early_stop=early, dims=(dimA, dimB))
    for _ in range(3):
        if not alt.sweep():
            break
    torch.cuda.synchronize()
    A, B = be.sess.get_factors()
    if rank == 0:
        np.savez(out, A=A, B=B)
    dist.barrier()
    be.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
