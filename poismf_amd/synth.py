"""Synthetic inputs for the BASELINE configs.  Host-side numpy only.

C1  README example (ref: README.md:91-100): 100 x 1000, 1e4 triplets, k = 5.
C2  uniform 1e5 x 1e5, 1e7 triplets, k = 50.
C3  uniform 1e6 x 1e5, 1e8 triplets, k = 50.   (C4 = C3 in fp32 on 8 GPUs)
C5  Last.FM-shaped 358 858 x 160 112, ~17 M nnz, power-law item degrees, k = 100.
"""
import numpy as np
import scipy.sparse as sp


def readme_coo():
    """The reference's README data, verbatim recipe (legacy numpy RandomState seed 1)."""
    rs = np.random.RandomState(1)  # == np.random.seed(1) followed by the module-level calls
    nusers, nitems, nobs = 10 ** 2, 10 ** 3, 10 ** 4
    user = rs.randint(nusers, size=nobs)
    item = rs.randint(nitems, size=nobs)
    count = 1 + rs.gamma(1, 1, size=nobs).astype(int)
    return sp.coo_matrix((count.astype(np.float64), (user, item)), shape=(nusers, nitems))


class Triplets:
    """COO triplets as plain arrays (row / col int64, data float64) with the few attributes of a SciPy COO matrix that
    poismf_amd.api reads -- at 1e8 triplets building the SciPy object only copies and re-casts the index arrays."""

    def __init__(self, row, col, data, shape):
        self.row, self.col, self.data, self.shape = row, col, data, tuple(shape)
        self.nnz = len(data)


def uniform_triplets(dimA, dimB, nnz, seed=1):
    """C2-C4: uniform random positions, values 1 + floor(Gamma(1,1)); duplicates are summed later
    by the CSR/CSC conversion exactly as SciPy does for the reference."""
    rng = np.random.default_rng(seed)
    row = rng.integers(0, dimA, nnz, dtype=np.int64)
    col = rng.integers(0, dimB, nnz, dtype=np.int64)
    val = rng.standard_gamma(1.0, nnz)
    np.floor(val, out=val)
    val += 1.0
    return Triplets(row, col, val, (dimA, dimB))


def _chunks(n, step):
    for lo in range(0, n, step):
        yield lo, min(n, lo + step)


def uniform_counts(dimA, dimB, nnz, seed=1, chunk=1 << 24):
    """Triplets per row and per column of uniform_triplets(dimA, dimB, nnz, seed), without keeping the triplets: the row / column
    streams are drawn chunk by chunk (same generator, same order, same values as the one-shot draw) and only counted.  Also returns the
    generator states at the start of the column stream and of the value stream, so that uniform_triplets_of can re-draw the three
    streams side by side."""
    rng = np.random.default_rng(seed)
    cntA, cntB = np.zeros(dimA, np.int64), np.zeros(dimB, np.int64)
    for lo, hi in _chunks(nnz, chunk):
        cntA += np.bincount(rng.integers(0, dimA, hi - lo, dtype=np.int64), minlength=dimA)
    st_col = rng.bit_generator.state
    for lo, hi in _chunks(nnz, chunk):
        cntB += np.bincount(rng.integers(0, dimB, hi - lo, dtype=np.int64), minlength=dimB)
    st_val = rng.bit_generator.state
    return cntA, cntB, (st_col, st_val)


def uniform_triplets_of(dimA, dimB, nnz, rowsA, rowsB, states, seed=1, chunk=1 << 24):
    """The triplets of uniform_triplets(dimA, dimB, nnz, seed) whose row lies in rowsA = (lo, hi) OR whose column lies in rowsB: what a
    rank that owns those row ranges of the two halves needs, and nothing else (one rank of eight keeps ~ 1/8 + 1/8 of the matrix instead
    of generating and holding all of it).  `states` comes from uniform_counts.  Bit-identical values: the three streams are re-drawn in
    lockstep from the generator states at which the one-shot draw starts each of them."""
    gens = [np.random.default_rng(seed), np.random.default_rng(seed), np.random.default_rng(seed)]
    gens[1].bit_generator.state = states[0]
    gens[2].bit_generator.state = states[1]
    keep_r, keep_c, keep_v = [], [], []
    for lo, hi in _chunks(nnz, chunk):
        r = gens[0].integers(0, dimA, hi - lo, dtype=np.int64)
        c = gens[1].integers(0, dimB, hi - lo, dtype=np.int64)
        v = gens[2].standard_gamma(1.0, hi - lo)
        m = ((r >= rowsA[0]) & (r < rowsA[1])) | ((c >= rowsB[0]) & (c < rowsB[1]))
        keep_r.append(r[m]); keep_c.append(c[m]); keep_v.append(v[m])
    val = np.concatenate(keep_v)
    np.floor(val, out=val)
    val += 1.0
    return Triplets(np.concatenate(keep_r), np.concatenate(keep_c), val, (dimA, dimB))


def uniform_coo(dimA, dimB, nnz, seed=1):
    """The same triplets as a SciPy COO matrix."""
    t = uniform_triplets(dimA, dimB, nnz, seed)
    return sp.coo_matrix((t.data, (t.row, t.col)), shape=t.shape)


def lastfm_like_coo(nusers=358858, nitems=160112, mean_deg=47, zipf_a=0.7, seed=1):
    """C5: per-user degree 1 + Poisson(mean_deg); items drawn from p_j ~ (j+1)^-a (duplicates within a
    user are summed by the conversion, which slightly lowers nnz); values 1 + floor(LogNormal(4, 1.3))."""
    rng = np.random.default_rng(seed)
    deg = np.minimum(nitems, 1 + rng.poisson(mean_deg, nusers)).astype(np.int64)
    nnz = int(deg.sum())
    row = np.repeat(np.arange(nusers, dtype=np.int64), deg)
    p = (np.arange(nitems, dtype=np.float64) + 1.0) ** (-zipf_a)
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    col = np.searchsorted(cdf, rng.random(nnz), side="right").astype(np.int64)
    np.minimum(col, nitems - 1, out=col)
    val = 1.0 + np.floor(rng.lognormal(4.0, 1.3, nnz))
    return sp.coo_matrix((val, (row, col)), shape=(nusers, nitems))


CONFIGS = {
    # name: (builder, k, method, use_float)
    "C1": (readme_coo, 5, "pg", True),
    "C2": (lambda: uniform_coo(10 ** 5, 10 ** 5, 10 ** 7), 50, "pg", True),
    "C3": (lambda: uniform_coo(10 ** 6, 10 ** 5, 10 ** 8), 50, "cg", False),
    "C4": (lambda: uniform_coo(10 ** 6, 10 ** 5, 10 ** 8), 50, "pg", True),
    "C5": (lastfm_like_coo, 100, "tncg", False),
}
