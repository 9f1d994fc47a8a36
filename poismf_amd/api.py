"""Python host side above the C-ABI (include/poismf_hip.h).

Mirrors the reference's binding layer for this path:

* ``_run_poismf``  -- same positional order, defaults, checks and exceptions as the Cython entry point
  (ref: poismf/poismf_c_wrapper.pxi:57-107; note *indices before indptr* here, the reverse of the C
  signature it forwards to);
* ``PoisMF``       -- the fit-path subset of the reference class (ref: poismf/__init__.py:205-232
  constructor, :336-374 ``fit``, :427-439 ``_fit``, :441-495 ``fit_unsafe``);
* ``Session``      -- the device-resident half-sweep API used by bench.py and the multi-GPU driver.

There is no CPU fallback anywhere in this module: if the HIP library cannot be loaded, or there is no
GPU, the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build
from . import harness

_METHOD = {"tncg": 1, "cg": 2, "pg": 3}  # ref: src/poismf.h:225
_LIBS = {}


class poismf_hip_params(C.Structure):
    pass


def _params_type(c_real):
    class Params(C.Structure):
        _fields_ = [("l2_reg", c_real), ("l1_reg", c_real), ("w_mult", c_real), ("step_size", c_real),
                    ("method", C.c_int), ("limit_step", C.c_int), ("maxupd", C.c_size_t),
                    ("early_stop", C.c_int), ("reuse_prev", C.c_int)]
    return Params


# every symbol include/poismf_hip.h declares
EXPORTED_SYMBOLS = (
    "run_poismf", "factors_multiple", "poismf_hip_coo_to_csr_csc", "predict_multiple", "topN", "poismf_hip_session_create", "poismf_hip_session_destroy", "poismf_hip_session_A",
    "poismf_hip_session_B", "poismf_hip_session_set_factors", "poismf_hip_session_get_factors",
    "poismf_hip_half_sweep", "poismf_hip_session_profile", "poismf_hip_session_kernel_time",
    "poismf_hip_session_nnz", "poismf_hip_selftest_log", "poismf_hip_session_eval_stats",
    "poismf_hip_session_create_coo", "poismf_hip_session_stream", "poismf_hip_session_factors_dirty", "poismf_hip_session_run",
    "poismf_hip_session_set_segments", "poismf_hip_session_segment_rows", "poismf_hip_half_sweep_segment", "poismf_hip_session_plan",
    "poismf_hip_session_launch_profile", "poismf_hip_session_decisions", "poismf_hip_session_decision_stats", "poismf_hip_factors_multiple_decisions",
    "poismf_hip_session_predict", "poismf_hip_session_topn", "poismf_hip_debug_row_eval", "poismf_hip_release_cache",
    "poismf_hip_set_device_cache_mb", "poismf_hip_session_colsum_blocks", "poismf_hip_session_colsum_partial", "poismf_hip_session_partials",
    "poismf_hip_session_partials_ready",
)


def load_library(use_float):
    """dlopen libpoismf_hip_{d,f}.so (built in-tree by poismf_amd.build) and declare its prototypes.  use_float = "r" loads
    libpoismf_hip_r.so, the reference's R ABI (int indices, double): same prototypes, index arrays are int32."""
    key = "r" if use_float == "r" else bool(use_float)
    if key in _LIBS:
        return _LIBS[key]
    path = _build.lib_path(key)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m poismf_amd.build` "
                           "(there is no CPU fallback for this path)")
    lib = C.CDLL(path)
    r = C.c_float if key is True else C.c_double
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    lib.run_poismf.argtypes = [vp] * 8 + [sz] * 3 + [r] * 4 + [i, C.c_bool, sz, sz] + [C.c_bool] * 3 + [i]
    lib.run_poismf.restype = i
    lib.factors_multiple.argtypes = [vp] * 7 + [i, sz, r, r, r, sz, sz, i, C.c_bool, C.c_bool, i]
    lib.factors_multiple.restype = i
    lib.poismf_hip_factors_multiple_decisions.argtypes = [vp] * 7 + [i, sz, r, r, r, sz, sz, i, C.c_bool, C.c_bool, vp]
    lib.poismf_hip_factors_multiple_decisions.restype = i
    lib.poismf_hip_session_decisions.argtypes = [vp, i, vp, sz]
    lib.poismf_hip_session_decisions.restype = i
    lib.poismf_hip_session_decision_stats.argtypes = [vp, i, C.POINTER(C.c_ulonglong)]
    lib.poismf_hip_session_decision_stats.restype = i
    lib.poismf_hip_release_cache.argtypes = []
    lib.poismf_hip_release_cache.restype = None
    lib.poismf_hip_set_device_cache_mb.argtypes = [sz]
    lib.poismf_hip_set_device_cache_mb.restype = sz
    lib.poismf_hip_session_colsum_blocks.argtypes = [vp, i]
    lib.poismf_hip_session_colsum_blocks.restype = i
    lib.poismf_hip_session_colsum_partial.argtypes = [vp, i, i, i]
    lib.poismf_hip_session_colsum_partial.restype = i
    lib.poismf_hip_session_partials.argtypes = [vp]
    lib.poismf_hip_session_partials.restype = vp
    lib.poismf_hip_session_partials_ready.argtypes = [vp]
    lib.poismf_hip_session_partials_ready.restype = None
    lib.predict_multiple.argtypes = [vp, vp, vp, vp, vp, sz, i, i]
    lib.predict_multiple.restype = None
    lib.topN.argtypes = [vp, vp, i, vp, sz, vp, sz, vp, vp, sz, sz, i]
    lib.topN.restype = i
    lib.poismf_hip_coo_to_csr_csc.argtypes = [vp, vp, vp, sz, sz, sz] + [vp] * 6 + [C.POINTER(sz)]
    lib.poismf_hip_coo_to_csr_csc.restype = i
    lib.poismf_hip_session_create.argtypes = [C.POINTER(vp), i, vp] + [vp] * 6 + [sz] * 3 + [sz] * 4
    lib.poismf_hip_session_create.restype = i
    lib.poismf_hip_session_create_coo.argtypes = [C.POINTER(vp), i, vp, vp, vp, vp, sz] + [sz] * 3 + [sz] * 4
    lib.poismf_hip_session_create_coo.restype = i
    lib.poismf_hip_session_destroy.argtypes = [vp]
    lib.poismf_hip_session_destroy.restype = None
    lib.poismf_hip_session_stream.argtypes = [vp]
    lib.poismf_hip_session_stream.restype = vp
    lib.poismf_hip_session_factors_dirty.argtypes = [vp, i]
    lib.poismf_hip_session_factors_dirty.restype = None
    for name in ("poismf_hip_session_A", "poismf_hip_session_B"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = vp
    for name in ("poismf_hip_session_set_factors", "poismf_hip_session_get_factors"):
        getattr(lib, name).argtypes = [vp, vp, vp]
        getattr(lib, name).restype = i
    lib.params_t = _params_type(r)
    lib.poismf_hip_half_sweep.argtypes = [vp, i, C.POINTER(lib.params_t), r, r, C.POINTER(sz)]
    lib.poismf_hip_half_sweep.restype = i
    lib.poismf_hip_session_set_segments.argtypes = [vp, i, i]
    lib.poismf_hip_session_set_segments.restype = i
    lib.poismf_hip_session_segment_rows.argtypes = [vp, i, i, C.POINTER(sz), C.POINTER(sz)]
    lib.poismf_hip_session_segment_rows.restype = i
    lib.poismf_hip_half_sweep_segment.argtypes = [vp, i, C.POINTER(lib.params_t), r, r, i, C.POINTER(sz)]
    lib.poismf_hip_half_sweep_segment.restype = i
    lib.poismf_hip_session_predict.argtypes = [vp, vp, vp, sz, vp]
    lib.poismf_hip_session_predict.restype = i
    lib.poismf_hip_session_topn.argtypes = [vp, sz, vp, sz, vp, sz, vp, vp, sz]
    lib.poismf_hip_session_topn.restype = i
    lib.poismf_hip_session_plan.argtypes = [vp, i, C.c_char_p, sz]
    lib.poismf_hip_session_plan.restype = sz
    lib.poismf_hip_session_launch_profile.argtypes = [vp, i, C.c_char_p, sz]
    lib.poismf_hip_session_launch_profile.restype = sz
    lib.poismf_hip_session_run.argtypes = [vp, C.POINTER(lib.params_t), sz, i]
    lib.poismf_hip_session_run.restype = i
    lib.poismf_hip_session_profile.argtypes = [vp, i]
    lib.poismf_hip_session_profile.restype = None
    lib.poismf_hip_session_kernel_time.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(sz)]
    lib.poismf_hip_session_kernel_time.restype = i
    lib.poismf_hip_session_nnz.argtypes = [vp, i]
    lib.poismf_hip_session_nnz.restype = sz
    lib.poismf_hip_selftest_log.argtypes = [sz, C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)]
    lib.poismf_hip_selftest_log.restype = i
    lib.poismf_hip_session_eval_stats.argtypes = [vp, i, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.poismf_hip_session_eval_stats.restype = i
    lib.real_t = r
    _LIBS[key] = lib
    return lib


def release_cache(use_float=None):
    """Hand the device arrays the library keeps from finished calls (only when the caller opted in, see set_device_cache_mb) back
    to the driver.  use_float = None: every flavour loaded in this process (each shared library keeps its own list)."""
    for key, lib in list(_LIBS.items()):
        if use_float is None or key == ("r" if use_float == "r" else bool(use_float)):
            lib.poismf_hip_release_cache()


def set_device_cache_mb(mb, use_float=None):
    """Opt in (mb > 0) to / out (0, the default) of keeping released device arrays for the next call: the reference frees everything
    before run_poismf returns (ref src/poismf.c:610-619) and so does this library unless told otherwise here or through
    POISMF_HIP_DEVICE_CACHE_MB.  Returns {flavour: previous limit in MB}.  use_float = None: every flavour ALREADY loaded in this process (as
    release_cache does; nothing is loaded -- or built -- for the sake of a limit); a named flavour is loaded if need be."""
    if use_float is None:
        return {k_: int(lib.poismf_hip_set_device_cache_mb(int(mb))) for k_, lib in list(_LIBS.items())}
    key = "r" if use_float == "r" else bool(use_float)
    return {key: int(load_library(key).poismf_hip_set_device_cache_mb(int(mb)))}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _check_arrays(use_float, reals, idx):
    dt = np.float32 if use_float else np.float64
    for a in reals:
        if a.dtype != dt or not a.flags["C_CONTIGUOUS"]:
            raise TypeError(f"expected C-contiguous {dt.__name__} arrays")
    for a in idx:
        if a.dtype != np.uint64 or not a.flags["C_CONTIGUOUS"]:
            raise TypeError("index arrays must be C-contiguous size_t (uint64)")


def _run_poismf(Xr, Xr_indices, Xr_indptr, Xc, Xc_indices, Xc_indptr, A, B, method="tncg", limit_step=0,
                l2_reg=1e9, l1_reg=0, w_mult=1., step_size=1e-7, niter=10, maxupd=1, early_stop=1,
                reuse_prev=1, handle_interrupt=1, nthreads=1):
    """Drop-in for c_funs_{float,double}._run_poismf (ref: poismf/poismf_c_wrapper.pxi:57-107).
    The precision is taken from A's dtype (the reference ships one module per precision)."""
    if Xr.shape[0] == 0:
        raise ValueError("'X' contains no non-zero entries.")                      # ref: pxi:74
    INT_MAX = np.iinfo(C.c_int).max
    if max(A.shape[0], A.shape[1], B.shape[0]) > INT_MAX:                           # ref: pxi:78-80
        raise ValueError("Error: integer overflow. Dimensions cannot be larger than 2^31-1.")
    use_float = A.dtype == np.float32
    _check_arrays(use_float, (Xr, Xc, A, B), (Xr_indices, Xr_indptr, Xc_indices, Xc_indptr))
    lib = load_library(use_float)
    c_method = _METHOD.get(method, 1)                                               # ref: pxi:85-91
    ret = lib.run_poismf(_ptr(A), _ptr(Xr), _ptr(Xr_indptr), _ptr(Xr_indices),
                         _ptr(B), _ptr(Xc), _ptr(Xc_indptr), _ptr(Xc_indices),
                         A.shape[0], B.shape[0], A.shape[1], l2_reg, l1_reg, w_mult, step_size, c_method,
                         bool(limit_step), int(niter), int(maxupd), bool(early_stop), bool(reuse_prev),
                         bool(handle_interrupt), int(nthreads))
    if ret == 1:
        raise MemoryError("Could not allocate enough memory.")                      # ref: pxi:104-105
    elif ret == 2 and not handle_interrupt:
        raise InterruptedError("Procedure was interrupted")                         # ref: pxi:106-107
    return ret


def _coo_arrays(coo, use_float):
    """(row, col, val) of a COO-like object as C-contiguous size_t / size_t / real_t arrays (no copy when they already are;
    non-negative int64 indices are reinterpreted in place)"""
    def ix(a):
        a = np.asarray(a)
        if a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]:
            return a.view(np.uint64)
        return np.ascontiguousarray(a, dtype=np.uint64)
    return ix(coo.row), ix(coo.col), np.ascontiguousarray(coo.data, dtype=np.float32 if use_float else np.float64)


def coo_to_csr_csc(coo, use_float):
    """GPU replacement of harness.process_data (ref: poismf/__init__.py:404-414): SciPy COO -> (csr, csc) tuples of
    (data real_t, indices size_t, indptr size_t) with duplicates summed and indices sorted."""
    lib = load_library(use_float)
    dt = np.float32 if use_float else np.float64
    n = coo.nnz
    if n == 0:
        raise ValueError("'X' contains no non-zero entries.")
    row, col, val = _coo_arrays(coo, use_float)
    dimA, dimB = coo.shape
    cv, ci, cp = np.empty(n, dt), np.empty(n, np.uint64), np.empty(dimA + 1, np.uint64)
    kv, ki, kp = np.empty(n, dt), np.empty(n, np.uint64), np.empty(dimB + 1, np.uint64)
    nnz = C.c_size_t(0)
    if lib.poismf_hip_coo_to_csr_csc(_ptr(row), _ptr(col), _ptr(val), n, dimA, dimB, _ptr(cv), _ptr(ci), _ptr(cp), _ptr(kv),
                                     _ptr(ki), _ptr(kp), C.byref(nnz)):
        raise MemoryError("Could not allocate enough memory.")
    m = nnz.value
    return (cv[:m].copy(), ci[:m].copy(), cp), (kv[:m].copy(), ki[:m].copy(), kp)


def _predict_multiple(out, A, B, ix_u, ix_i, nthreads=1):
    """Drop-in for c_funs._predict_multiple (ref: poismf/poismf_c_wrapper.pxi:109-112): out[i] = A[ix_u[i]] . B[ix_i[i]]."""
    use_float = A.dtype == np.float32
    _check_arrays(use_float, (out, A, B), (ix_u, ix_i))
    load_library(use_float).predict_multiple(_ptr(out), _ptr(A), _ptr(B), _ptr(ix_u), _ptr(ix_i), ix_u.shape[0], A.shape[1],
                                             int(nthreads))


def _call_topN(a_vec, B, include_ix, exclude_ix, top_n=10, output_score=0, nthreads=1):
    """Drop-in for c_funs._call_topN (ref: poismf/poismf_c_wrapper.pxi:207-246): returns (indices, scores)."""
    use_float = B.dtype == np.float32
    _check_arrays(use_float, (a_vec, B), (include_ix, exclude_ix))
    dt = B.dtype
    n_inc = include_ix.shape[0]
    n_exc = 0 if n_inc else exclude_ix.shape[0]
    outp_ix = np.empty(top_n, dtype=np.uint64)
    outp_score = np.empty(top_n if output_score else 0, dtype=dt)
    rc = load_library(use_float).topN(_ptr(a_vec), _ptr(B), B.shape[1], _ptr(include_ix) if n_inc else None, n_inc,
                                      _ptr(exclude_ix) if n_exc else None, n_exc, _ptr(outp_ix),
                                      _ptr(outp_score) if output_score else None, int(top_n), B.shape[0], int(nthreads))
    if rc == 1:
        raise MemoryError("Could not allocate enough memory.")
    if rc == 2:
        raise ValueError("invalid combination of include / exclude / top_n")
    return outp_ix, outp_score


def _predict_factors_multiple(B, Bsum, Amean, Xr_indptr, Xr_indices, Xr, l2_reg=1e9, w_mult=1., step_size=1e-7,
                              niter=10, maxupd=1, method="tncg", limit_step=0, reuse_mean=1, nthreads=1):
    """Drop-in for c_funs_{float,double}._predict_factors_multiple (ref: poismf/poismf_c_wrapper.pxi:147-199):
    same positional order and defaults; returns the new factors A [n_new x k]."""
    use_float = B.dtype == np.float32
    _check_arrays(use_float, (B, Bsum, Amean, Xr), (Xr_indptr, Xr_indices))
    lib = load_library(use_float)
    k = B.shape[1]
    dimA = Xr_indptr.shape[0] - 1
    A = np.empty((dimA, k), dtype=B.dtype)
    ret = lib.factors_multiple(_ptr(A), _ptr(B), _ptr(Bsum), _ptr(Amean), _ptr(Xr) if Xr.shape[0] else None, _ptr(Xr_indptr),
                               _ptr(Xr_indices) if Xr_indices.shape[0] else None, k, dimA, l2_reg, w_mult, step_size,
                               int(niter), int(maxupd), _METHOD.get(method, 1), bool(limit_step), bool(reuse_mean),
                               int(nthreads))
    if ret:
        raise MemoryError("Could not allocate enough memory.")                      # ref: pxi:205-206
    return A


def factors_multiple_with_decisions(B, Bsum, Amean, Xr_indptr, Xr_indices, Xr, l2_reg=1e9, w_mult=1., step_size=1e-7,
                                    niter=10, maxupd=1, method="tncg", limit_step=0, reuse_mean=1):
    """_predict_factors_multiple plus what each row's solver decided: returns (A, iterations, evaluations, rc) -- the
    numbers the reference's minimize_nonneg_cg / tnc hand back (testing aid, include/poismf_hip.h)."""
    use_float = B.dtype == np.float32
    _check_arrays(use_float, (B, Bsum, Amean, Xr), (Xr_indptr, Xr_indices))
    lib = load_library(use_float)
    k = B.shape[1]
    dimA = Xr_indptr.shape[0] - 1
    A = np.empty((dimA, k), dtype=B.dtype)
    dec = np.zeros((dimA, 2), dtype=np.uint32)
    ret = lib.poismf_hip_factors_multiple_decisions(_ptr(A), _ptr(B), _ptr(Bsum), _ptr(Amean), _ptr(Xr) if Xr.shape[0] else None,
                                                    _ptr(Xr_indptr), _ptr(Xr_indices) if Xr_indices.shape[0] else None, k, dimA, l2_reg,
                                                    w_mult, step_size, int(niter), int(maxupd), _METHOD.get(method, 1), bool(limit_step),
                                                    bool(reuse_mean), _ptr(dec))
    if ret:
        raise MemoryError("Could not allocate enough memory.")
    return A, (dec[:, 0] & 0xffffff).astype(np.int64), dec[:, 1].astype(np.int64), (dec[:, 0] >> 24).astype(np.int64)


def debug_row_eval(B, Bsum, point, Xr_indptr, Xr_indices, Xr, l2_reg, w_mult=1., which=0):
    """Testing aid (include/poismf_hip.h, poismf_hip_debug_row_eval): the device's fun_single + grad_single (which = 0) or fun_and_grad
    (which = 1) at `point` for every row of the CSR; returns (f [n] float64, G [n x k])."""
    use_float = B.dtype == np.float32
    _check_arrays(use_float, (B, Bsum, point, Xr), (Xr_indptr, Xr_indices))
    lib = load_library(use_float)
    k = B.shape[1]
    n = Xr_indptr.shape[0] - 1
    G = np.empty((n, k), dtype=B.dtype)
    f = np.empty(n, dtype=np.float64)
    real = C.c_float if use_float else C.c_double
    fn = lib.poismf_hip_debug_row_eval
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_size_t, real, real, C.c_int]
    ret = fn(_ptr(G), _ptr(f), _ptr(B), _ptr(Bsum), _ptr(point), _ptr(Xr), _ptr(Xr_indptr), _ptr(Xr_indices), k, n, l2_reg, w_mult, int(which))
    if ret:
        raise MemoryError("poismf_hip_debug_row_eval failed")
    return f, G


class PoisMF:
    """Fit-path subset of the reference's PoisMF (ref: poismf/__init__.py:205-495).  Only what sits on
    the factor-update path is mirrored: constructor arguments that reach run_poismf, ``fit`` for SciPy
    COO input, ``fit_unsafe``, and the post-fit ``Bsum`` / ``Amean``."""

    def __init__(self, k=50, method="tncg", l2_reg="auto", l1_reg=0.0, niter="auto", maxupd="auto",
                 limit_step=True, initial_step=1e-7, early_stop=True, reuse_prev=False, weight_mult=1.0,
                 random_state=1, use_float=True, handle_interrupt=True, nthreads=-1):
        assert method in ("tncg", "cg", "pg")
        self.k = int(k)
        self.method = method
        self.l2_reg_, self.maxupd_, self.niter_ = harness.auto_defaults(method, self.k, l2_reg, maxupd, niter)
        assert self.k > 0 and self.niter_ >= 1 and self.maxupd_ >= 1
        assert self.l2_reg_ >= 0. and l1_reg >= 0. and initial_step > 0. and weight_mult > 0.
        self.l1_reg_ = float(l1_reg)
        self.limit_step = bool(limit_step)
        self.initial_step = float(initial_step)
        self.early_stop = bool(early_stop)
        self.reuse_prev = bool(reuse_prev)
        self.weight_mult = float(weight_mult)
        self.random_state = random_state
        self.use_float = bool(use_float)
        self.handle_interrupt = bool(handle_interrupt)
        self.nthreads_ = 1 if nthreads < 1 else int(nthreads)
        self.is_fitted = False

    def fit(self, X):
        # COO -> CSR + CSC on the device (bit-identical to the SciPy conversion of the reference's _process_data);
        # the converted matrix stays in HBM and the alternation runs on it in place (run_poismf's loop, same return
        # codes and exceptions) -- nothing but the triplets goes up and nothing but the factors comes back
        import scipy.sparse as sp
        self.nusers, self.nitems = X.shape
        self.A, self.B = harness.initialize_matrices(self.nusers, self.nitems, self.k, self.use_float,
                                                     self.random_state)
        sess = Session.from_coo(sp.coo_matrix(X), self.k, self.use_float)
        try:
            sess.set_factors(self.A, self.B)
            p = sess.make_params(self.method, self.l2_reg_, self.l1_reg_, self.weight_mult, self.initial_step, self.limit_step,
                                 self.maxupd_, self.early_stop, self.reuse_prev)
            ret = sess.run(p, self.niter_, self.handle_interrupt)
            A, B = sess.get_factors()
            self.A[...] = A
            self.B[...] = B
        finally:
            sess.close()
        if ret == 1:
            raise MemoryError("Could not allocate enough memory.")
        elif ret == 2 and not self.handle_interrupt:
            raise InterruptedError("Procedure was interrupted")
        self.Bsum = self.B.sum(axis=0) + self.l1_reg_
        self.Amean = self.A.mean(axis=0)
        self.is_fitted = True
        return self

    def fit_unsafe(self, A, B, Xcsr, Xcsc):
        self.A, self.B = A, B
        self.nusers, self.nitems = A.shape[0], B.shape[0]
        self._fit((Xcsr.data, Xcsr.indices, Xcsr.indptr), (Xcsc.data, Xcsc.indices, Xcsc.indptr))
        self.is_fitted = True
        return self

    def _fit(self, csr, csc):                                                        # ref: __init__.py:427-439
        _run_poismf(csr[0], csr[1], csr[2], csc[0], csc[1], csc[2], self.A, self.B, self.method,
                    self.limit_step, self.l2_reg_, self.l1_reg_, self.weight_mult, self.initial_step,
                    self.niter_, self.maxupd_, self.early_stop, self.reuse_prev, self.handle_interrupt,
                    self.nthreads_)
        self.Bsum = self.B.sum(axis=0) + self.l1_reg_
        self.Amean = self.A.mean(axis=0)


def _transform(self, X):
    """Factors for new rows given as a SciPy CSR / COO matrix (ref: poismf/__init__.py:619-692, transform)."""
    import scipy.sparse as sp
    assert self.is_fitted and X.shape[0] > 0
    csr = sp.csr_matrix(X)
    csr.sum_duplicates(); csr.sort_indices()
    dt = np.float32 if self.use_float else np.float64
    return _predict_factors_multiple(
        self.B, self.Bsum.astype(dt), self.Amean.astype(dt), csr.indptr.astype(np.uint64), csr.indices.astype(np.uint64),
        csr.data.astype(dt), self.l2_reg_, self.weight_mult, self.initial_step, self.niter_, self.maxupd_, self.method,
        self.limit_step, self.reuse_prev, self.nthreads_)


PoisMF.transform = _transform


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can alias session-owned device memory."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class Session:
    """Device-resident half-sweeps (include/poismf_hip.h section 2).  X, A and B stay in HBM; each call
    runs one half of the alternation on this session's row shard."""

    def __init__(self, csr, csc, dimA, dimB, k, use_float, device=0, stream=None, shardA=None, shardB=None, _coo=None):
        self.lib = load_library(use_float)
        self.use_float = bool(use_float)
        self.dimA, self.dimB, self.k = int(dimA), int(dimB), int(k)
        self.shardA = tuple(shardA) if shardA is not None else (0, self.dimA)
        self.shardB = tuple(shardB) if shardB is not None else (0, self.dimB)
        h = C.c_void_p()
        if _coo is not None:
            row, col, val = _coo
            rc = self.lib.poismf_hip_session_create_coo(
                C.byref(h), int(device), C.c_void_p(stream or 0), _ptr(row), _ptr(col), _ptr(val), len(val),
                self.dimA, self.dimB, self.k, self.shardA[0], self.shardA[1], self.shardB[0], self.shardB[1])
            if rc == 3:
                raise ValueError("a row / column index of the triplets lies outside the matrix")
        else:
            _check_arrays(use_float, (csr[0], csc[0]), (csr[1], csr[2], csc[1], csc[2]))
            if len(csr[0]) == 0:
                raise ValueError("'X' contains no non-zero entries.")
            rc = self.lib.poismf_hip_session_create(
                C.byref(h), int(device), C.c_void_p(stream or 0), _ptr(csr[0]), _ptr(csr[2]), _ptr(csr[1]),
                _ptr(csc[0]), _ptr(csc[2]), _ptr(csc[1]), self.dimA, self.dimB, self.k,
                self.shardA[0], self.shardA[1], self.shardB[0], self.shardB[1])
        if rc != 0 or not h.value:
            raise MemoryError("poismf_hip_session_create failed (no usable HIP device or out of memory)")
        self.h = h

    @classmethod
    def from_coo(cls, coo, k, use_float, device=0, stream=None, shardA=None, shardB=None):
        """Session whose CSR and CSC are built on the device from the triplets of a SciPy COO matrix (duplicates summed,
        indices sorted: bit-identical to tocsr() / tocsc()); with shards, only the triplets of the shard's rows
        (CSR) / columns (CSC) are kept."""
        if coo.nnz == 0:
            raise ValueError("'X' contains no non-zero entries.")
        return cls(None, None, coo.shape[0], coo.shape[1], k, use_float, device, stream, shardA, shardB,
                   _coo=_coo_arrays(coo, use_float))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.poismf_hip_session_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def set_factors(self, A, B):
        _check_arrays(self.use_float, (A, B), ())
        assert A.shape == (self.dimA, self.k) and B.shape == (self.dimB, self.k)
        if self.lib.poismf_hip_session_set_factors(self.h, _ptr(A), _ptr(B)):
            raise RuntimeError("poismf_hip_session_set_factors failed")

    def get_factors(self, out=None):
        """(A, B) on the host; out = (A, B): into these C-contiguous arrays of the session's shapes and precision."""
        dt = np.float32 if self.use_float else np.float64
        if out is not None:
            A, B = out
            for M, n in ((A, self.dimA), (B, self.dimB)):
                if M.dtype != dt or M.shape != (n, self.k) or not M.flags.c_contiguous:
                    raise ValueError("get_factors: out arrays must be C-contiguous, of the session's shapes and precision")
        else:
            A = np.empty((self.dimA, self.k), dt)
            B = np.empty((self.dimB, self.k), dt)
        if self.lib.poismf_hip_session_get_factors(self.h, _ptr(A), _ptr(B)):
            raise RuntimeError("poismf_hip_session_get_factors failed")
        return A, B

    def device_arrays(self):
        """(A, B) as objects exposing __cuda_array_interface__ over the session's device buffers."""
        ts = "<f4" if self.use_float else "<f8"
        return (_DevArray(self.lib.poismf_hip_session_A(self.h), (self.dimA, self.k), ts),
                _DevArray(self.lib.poismf_hip_session_B(self.h), (self.dimB, self.k), ts))

    def make_params(self, method, l2_reg, l1_reg=0.0, w_mult=1.0, step_size=1e-7, limit_step=True, maxupd=1,
                    early_stop=False, reuse_prev=False):
        return self.lib.params_t(l2_reg, l1_reg, w_mult, step_size, _METHOD[method], int(limit_step), int(maxupd),
                                 int(early_stop), int(reuse_prev))

    def half_sweep(self, which, params, step_size, cnst_div, want_unchanged=False, seg=None):
        """One half-sweep over the session's shard, or (seg = j) over its segment j only."""
        n = C.c_size_t(0)
        if seg is None:
            rc = self.lib.poismf_hip_half_sweep(self.h, int(which), C.byref(params), step_size, cnst_div,
                                                C.byref(n) if want_unchanged else None)
        else:
            rc = self.lib.poismf_hip_half_sweep_segment(self.h, int(which), C.byref(params), step_size, cnst_div, int(seg),
                                                        C.byref(n) if want_unchanged else None)
        if rc:
            raise RuntimeError("poismf_hip_half_sweep failed")
        return n.value

    def set_segments(self, which, nseg):
        n = self.lib.poismf_hip_session_set_segments(self.h, int(which), int(nseg))
        if n < 1:
            raise RuntimeError("poismf_hip_session_set_segments failed")
        return n

    def segment_rows(self, which, seg):
        b, e = C.c_size_t(0), C.c_size_t(0)
        if self.lib.poismf_hip_session_segment_rows(self.h, int(which), int(seg), C.byref(b), C.byref(e)):
            raise IndexError("no such segment")
        return b.value, e.value

    def real(self, v):
        """v rounded to the session's real_t, as a Python float (run_poismf does its step / divisor arithmetic in
        real_t, ref: src/poismf.c:511, :532)"""
        return float(np.float32(v)) if self.use_float else float(v)

    def cnst_div(self, l2_reg, step_size):
        """The PG divisor 1 / (1 + 2 l2 step) evaluated as run_poismf evaluates it (ref: src/poismf.c:511): l2 and the
        step are real_t, the expression is double, the result is stored as real_t."""
        return self.real(1. / (1. + 2. * self.real(l2_reg) * self.real(step_size)))

    def sweep(self, params, step_size):
        """One full outer iteration with the reference's schedule (ref: src/poismf.c:506-608): B half, then
        (PG) halve the step, then A half.  Returns the step for the next iteration."""
        step_size = self.real(step_size)
        cnst_div = self.cnst_div(params.l2_reg, step_size)
        self.half_sweep(0, params, step_size, cnst_div)
        if params.method == _METHOD["pg"]:
            step_size = self.real(step_size * 0.5)
        self.half_sweep(1, params, step_size, cnst_div)
        return step_size

    def run(self, params, niter, handle_interrupt=True):
        """run_poismf's whole loop on this session (return codes 0 / 1 / 2 as run_poismf)"""
        return self.lib.poismf_hip_session_run(self.h, C.byref(params), int(niter), int(bool(handle_interrupt)))

    def stream(self):
        """the hipStream_t this session enqueues on, as an integer handle"""
        return self.lib.poismf_hip_session_stream(self.h) or 0

    def factors_dirty(self, which):
        """tell the session that factor `which` (0: B, 1: A) was written through a device pointer obtained earlier"""
        self.lib.poismf_hip_session_factors_dirty(self.h, int(which))

    def colsum_blocks(self, which):
        """blocks the column sums over the FIXED factor of half `which` are cut into (include/poismf_hip.h)"""
        return int(self.lib.poismf_hip_session_colsum_blocks(self.h, int(which)))

    def colsum_partial(self, which, b_lo, b_hi):
        if self.lib.poismf_hip_session_colsum_partial(self.h, int(which), int(b_lo), int(b_hi)):
            raise RuntimeError("poismf_hip_session_colsum_partial failed")

    def partials_array(self, which):
        """the session's [blocks x k] partial sums of half `which`, as a __cuda_array_interface__ object aliasing device memory"""
        nb = self.colsum_blocks(which)
        return _DevArray(self.lib.poismf_hip_session_partials(self.h), (nb, self.k), "<f4" if self.use_float else "<f8")

    def partials_ready(self):
        self.lib.poismf_hip_session_partials_ready(self.h)

    def profile(self, enable=True):
        self.lib.poismf_hip_session_profile(self.h, int(enable))

    def kernel_time(self, which):
        ms, n = C.c_double(0), C.c_size_t(0)
        if self.lib.poismf_hip_session_kernel_time(self.h, int(which), C.byref(ms), C.byref(n)):
            raise RuntimeError("poismf_hip_session_kernel_time failed")
        return ms.value, n.value

    def eval_stats(self, which):
        """(tile passes, sum over rows of passes x nonzeros) of this half since profile(True)"""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        if self.lib.poismf_hip_session_eval_stats(self.h, int(which), C.byref(a), C.byref(b)):
            raise RuntimeError("poismf_hip_session_eval_stats failed")
        return a.value, b.value

    def decisions(self, which):
        """(iterations, evaluations, rc) per row of this session's shard of half `which` (0: B rows, 1: A rows) in the most recent
        half-sweep since profile(True) -- what the reference's minimize_nonneg_cg / tnc return and its drivers drop"""
        lo, hi = self.shardA if which else self.shardB
        dec = np.zeros((hi - lo, 2), dtype=np.uint32)
        if self.lib.poismf_hip_session_decisions(self.h, int(which), _ptr(dec), hi - lo):
            raise RuntimeError("poismf_hip_session_decisions: profiling is off")
        return (dec[:, 0] & 0xffffff).astype(np.int64), dec[:, 1].astype(np.int64), (dec[:, 0] >> 24).astype(np.int64)

    def decision_stats(self, which):
        """sums over this shard's rows of half `which` in the most recent half-sweep since profile(True):
        dict(iterations, evaluations, nnz_iterations, nnz_evaluations)"""
        out = (C.c_ulonglong * 4)()
        if self.lib.poismf_hip_session_decision_stats(self.h, int(which), out):
            raise RuntimeError("poismf_hip_session_decision_stats: profiling is off")
        return dict(iterations=out[0], evaluations=out[1], nnz_iterations=out[2], nnz_evaluations=out[3])

    def nnz(self, which):
        return self.lib.poismf_hip_session_nnz(self.h, int(which))

    def predict(self, ix_u, ix_i):
        """A[ix_u[i]] . B[ix_i[i]] from the resident factors (ref: src/pred.c:42-64)"""
        ix_u = np.ascontiguousarray(ix_u, dtype=np.uint64)
        ix_i = np.ascontiguousarray(ix_i, dtype=np.uint64)
        out = np.empty(len(ix_u), np.float32 if self.use_float else np.float64)
        rc = self.lib.poismf_hip_session_predict(self.h, _ptr(ix_u), _ptr(ix_i), len(ix_u), _ptr(out))
        if rc == 2:
            raise IndexError("user / item index out of range")
        if rc:
            raise MemoryError("poismf_hip_session_predict failed")
        return out

    def topn(self, user, top_n=10, include_ix=(), exclude_ix=(), output_score=False):
        """top-N items of user `user` (row of the resident A) by score, descending (ref: src/topN.c:112-284)"""
        inc = np.ascontiguousarray(include_ix, dtype=np.uint64)
        exc = np.ascontiguousarray(exclude_ix, dtype=np.uint64)
        ix = np.empty(top_n, np.uint64)
        sc = np.empty(top_n if output_score else 0, np.float32 if self.use_float else np.float64)
        rc = self.lib.poismf_hip_session_topn(self.h, int(user), _ptr(inc) if len(inc) else None, len(inc), _ptr(exc) if len(exc) else None,
                                              len(exc), _ptr(ix), _ptr(sc) if output_score else None, int(top_n))
        if rc == 2:
            raise ValueError("invalid combination of include / exclude / top_n, or an index out of range")
        if rc:
            raise MemoryError("poismf_hip_session_topn failed")
        return ix, sc

    def _text(self, fn, which):
        """a text report of the library, whole: ask for the length first (a fixed buffer cut long plans mid-item)"""
        n = fn(self.h, int(which), None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        fn(self.h, int(which), buf, len(buf))
        return [item.strip() for item in buf.value.decode().split(";") if item.strip()]

    def plan(self, which):
        """the launches of the most recent half-sweep of half `which`: [(kernel instance, rows), ...]"""
        out = []
        for item in self._text(self.lib.poismf_hip_session_plan, which):
            name, rows = item.rsplit(" rows=", 1)
            out.append((name.strip(), int(rows)))
        return out

    def launch_profile(self, which):
        """per-launch timings of half `which` since profile(True): [dict(kernel, rows, nnz, calls, ms)], ms summed over calls"""
        out = []
        for item in self._text(self.lib.poismf_hip_session_launch_profile, which):
            name, rest = item.split(" rows=", 1)
            f = dict(kv.split("=") for kv in ("rows=" + rest).split())
            out.append(dict(kernel=name.strip(), rows=int(f["rows"]), nnz=int(f["nnz"]), calls=int(f["calls"]), ms=float(f["ms"])))
        return out
