"""Host-side mirror of the part of the reference's Python class that sits on the fit path
(SURVEY.md section 8a row H): data coercion, initialisation and hyper-parameter defaults.

ref: poismf/__init__.py:250-255 (auto defaults), :376-416 (_process_data), :419-425
(_initialize_matrices).  Nothing here touches the GPU; poismf_amd.api does.
"""
import ctypes

import numpy as np
import scipy.sparse as sp


def auto_defaults(method, k, l2_reg="auto", maxupd="auto", niter="auto"):
    """ref: poismf/__init__.py:250-255.  (R differs: pg maxupd auto = 1, ref: R/poismf.R:241.)"""
    assert method in ("tncg", "cg", "pg")
    if l2_reg == "auto":
        l2_reg = {"tncg": 1e3, "cg": 1e4, "pg": 1e9}[method]
    if maxupd == "auto":
        maxupd = {"tncg": 15 * k, "cg": 5, "pg": 10}[method]
    if niter == "auto":
        niter = {"tncg": 10, "cg": 30, "pg": 10}[method]
    return float(l2_reg), int(maxupd), int(niter)


def process_data(coo, use_float):
    """COO -> (CSR, CSC) with duplicates summed and per-row indices sorted (SciPy tocsr/tocsc), values
    as real_t, indices/indptr as size_t, C-contiguous.  ref: poismf/__init__.py:404-414."""
    dtype = ctypes.c_float if use_float else ctypes.c_double
    coo = sp.coo_matrix(coo)
    out = []
    for m in (coo.tocsr(), coo.tocsc()):
        m.sum_duplicates()
        m.sort_indices()
        indices = np.require(m.indices, dtype=ctypes.c_size_t, requirements=["ENSUREARRAY", "C_CONTIGUOUS"])
        indptr = np.require(m.indptr, dtype=ctypes.c_size_t, requirements=["ENSUREARRAY", "C_CONTIGUOUS"])
        data = np.require(m.data, dtype=dtype, requirements=["ENSUREARRAY", "C_CONTIGUOUS"])
        out.append((data, indices, indptr))
    return out[0], out[1]


def initialize_matrices(nusers, nitems, k, use_float, random_state=1):
    """A, B = 0.3 + U[0, 0.01), A drawn first, from numpy's default_rng, in fp64 then cast.
    ref: poismf/__init__.py:419-425."""
    rng = random_state if isinstance(random_state, np.random.Generator) else np.random.default_rng(int(random_state))
    A = 0.3 + rng.uniform(low=0, high=0.01, size=(nusers, k))
    B = 0.3 + rng.uniform(low=0, high=0.01, size=(nitems, k))
    if use_float:
        A = A.astype(np.float32)
        B = B.astype(np.float32)
    return np.ascontiguousarray(A), np.ascontiguousarray(B)


def poisson_objective(A, B, csr, l2_reg, l1_reg=0.0, w_mult=1.0):
    """Regularised negative Poisson log-likelihood in fp64:
    sum_c colsum(A)_c colsum(B)_c - w sum_nz x log(a.b) + l2 (|A|^2 + |B|^2) + l1 (|A|_1 + |B|_1)."""
    data, indices, indptr = csr
    A64 = np.asarray(A, np.float64)
    B64 = np.asarray(B, np.float64)
    rows = np.repeat(np.arange(A64.shape[0]), np.diff(indptr.astype(np.int64)))
    pred = np.einsum("ij,ij->i", A64[rows], B64[indices.astype(np.int64)])
    with np.errstate(divide="ignore", invalid="ignore"):
        ll = float(np.sum(np.asarray(data, np.float64) * np.log(pred)))
    dense = float(A64.sum(0) @ B64.sum(0))
    reg = l2_reg * (float((A64 ** 2).sum()) + float((B64 ** 2).sum()))
    reg += l1_reg * (float(A64.sum()) + float(B64.sum()))
    return dense - w_mult * ll + reg
