"""bench.py's multi-GPU input builder: the per-rank CSR / CSC shards must tile the CSR / CSC of the stacked row
blocks exactly (small sizes, CPU only)."""
import importlib.util
import os

import numpy as np
import scipy.sparse as sp

from poismf_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_shards_tile_the_stacked_matrix():
    b = _bench()
    world, rows, dimB, nnz = 4, 300, 240, 5000
    blocks = [sp.csr_matrix(synth.uniform_coo(rows, dimB, nnz, seed=1 + r)) for r in range(world)]
    full = sp.vstack(blocks).tocsr()
    full.sum_duplicates(); full.sort_indices()
    fcsc = full.tocsc(); fcsc.sort_indices()
    total = 0
    for rank in range(world):
        csr, csc, dimA, dimB2, rA, rB = b.build_inputs(rank, world, False, BLOCK_ROWS=rows, DIMB=dimB, BLOCK_NNZ=nnz)
        assert (dimA, dimB2) == (rows * world, dimB) and rA[rank] == (rank * rows, (rank + 1) * rows)
        a0, a1 = rA[rank]
        p = csr[2].astype(np.int64)
        assert p[a0] == 0 and p[-1] == len(csr[0]) and np.all(p[:a0] == 0) and np.all(p[a1:] == p[a1])
        ref = full[a0:a1]
        assert np.array_equal(p[a0:a1 + 1], ref.indptr) and np.array_equal(csr[1], ref.indices) and np.array_equal(csr[0], ref.data)
        c0, c1 = rB[rank]
        q = csc[2].astype(np.int64)
        refc = fcsc[:, c0:c1]
        assert np.array_equal(q[c0:c1 + 1], refc.indptr) and np.array_equal(csc[1], refc.indices) and np.array_equal(csc[0], refc.data)
        total += len(csr[0])
    assert total == full.nnz
