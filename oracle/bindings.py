"""ctypes bindings for the CPU oracle (oracle/liboracle_{d,f}.so) and, when present, the compiled
reference (oracle/_ref/libpoismf_ref_{d,f}.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, scripts/make_golden.py, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never from the poismf_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

METHOD = {"tncg": 1, "cg": 2, "pg": 3}  # ref: src/poismf.h:225


def build(ref=False):
    """Compile the oracle (and optionally the reference, only possible where /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _dt(is_float):
    return (np.float32, C.c_float) if is_float else (np.float64, C.c_double)


def _p(arr):
    return arr.ctypes.data_as(C.c_void_p) if arr is not None else None


class _Lib:
    def __init__(self, path, is_float):
        self.lib = C.CDLL(path)
        self.is_float = is_float
        self.np_t, self.c_t = _dt(is_float)

    def real(self, x):
        return np.ascontiguousarray(x, dtype=self.np_t)

    @staticmethod
    def ix(x):
        return np.ascontiguousarray(x, dtype=np.uint64)


class Oracle(_Lib):
    """This repo's C restatement."""

    def __init__(self, is_float=False, blas_flavour=False, fma_axpy=False):
        """blas_flavour=True loads oracle/_ref/liboracle_blas_*.so: the same restatement with its k-length
        sums routed through the reference's own BLAS (only exists next to the compiled reference).
        fma_axpy=True loads liboracle_fma_*.so: y += a x rounded once per element, as a BLAS with fused
        multiply-add (and the GPU) computes it -- see vaxpy in poismf_oracle.c for why that matters in fp32 CG."""
        if blas_flavour:
            path = os.path.join(HERE, "_ref", "liboracle_blas_f.so" if is_float else "liboracle_blas_d.so")
        elif fma_axpy:
            path = os.path.join(HERE, "liboracle_fma_f.so" if is_float else "liboracle_fma_d.so")
            if not os.path.exists(path):
                build()
        else:
            path = os.path.join(HERE, "liboracle_f.so" if is_float else "liboracle_d.so")
            if not os.path.exists(path):
                build()
        super().__init__(path, is_float)
        L, r = self.lib, self.c_t
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        L.oracle_calc_grad_pgd.argtypes = [vp, vp, vp, vp, vp, sz, i]
        L.oracle_calc_grad_pgd.restype = None
        L.oracle_calc_fun_single.argtypes = [vp, vp, vp, vp, vp, sz, i, r, r]
        L.oracle_calc_fun_single.restype = r
        L.oracle_calc_grad_single.argtypes = [vp, vp, vp, vp, vp, vp, sz, i, r, r, i]
        L.oracle_calc_grad_single.restype = None
        L.oracle_calc_fun_and_grad.argtypes = [vp, vp, vp, vp, vp, vp, sz, i, r, r]
        L.oracle_calc_fun_and_grad.restype = r
        L.oracle_cg_row.argtypes = [vp, vp, vp, vp, vp, sz, i, r, r, sz, i, vp, vp, vp]
        L.oracle_cg_row.restype = i
        L.oracle_tnc_row.argtypes = [vp, vp, vp, vp, vp, sz, i, r, r, i, vp, vp, vp]
        L.oracle_tnc_row.restype = i
        L.oracle_sum_by_cols.argtypes = [vp, vp, sz, sz]
        L.oracle_sum_by_cols.restype = None
        L.oracle_pg_iteration.argtypes = [vp, vp, vp, vp, vp, sz, sz, r, vp, vp, r, r, sz, i]
        L.oracle_pg_iteration.restype = None
        L.oracle_cg_iteration.argtypes = [vp, vp, vp, vp, vp, sz, sz, i, vp, r, r, sz, vp, i]
        L.oracle_cg_iteration.restype = None
        L.oracle_tncg_iteration.argtypes = [vp, vp, i, vp, vp, vp, sz, sz, vp, r, r, i, i, vp, i]
        L.oracle_tncg_iteration.restype = i
        L.oracle_run_poismf.argtypes = [vp] * 8 + [sz] * 3 + [r] * 4 + [i, C.c_bool, sz, sz] + [C.c_bool] * 3 + [i]
        L.oracle_run_poismf.restype = i
        L.oracle_factors_multiple.argtypes = [vp] * 7 + [i, sz, r, r, r, sz, sz, i, C.c_bool, C.c_bool, i]
        L.oracle_factors_multiple.restype = i
        L.oracle_predict_multiple.argtypes = [vp, vp, vp, vp, vp, sz, i, i]
        L.oracle_predict_multiple.restype = None
        L.oracle_topn.argtypes = [vp, vp, i, vp, sz, vp, sz, vp, vp, sz, sz]
        L.oracle_topn.restype = i

    # ---- G1 primitives -------------------------------------------------------------------
    def calc_grad_pgd(self, a, F, xval, xind):
        k = F.shape[1]
        out = np.empty(k, self.np_t)
        self.lib.oracle_calc_grad_pgd(_p(out), _p(a), _p(F), _p(xval), _p(xind), len(xval), k)
        return out

    def calc_fun_single(self, a, F, bsum, xval, xind, l2, w):
        return self.lib.oracle_calc_fun_single(_p(a), _p(F), _p(bsum), _p(xval), _p(xind), len(xval),
                                               F.shape[1], l2, w)

    def calc_grad_single(self, a, F, bsum, xval, xind, l2, w, weighted=False):
        g = np.empty(F.shape[1], self.np_t)
        self.lib.oracle_calc_grad_single(_p(g), _p(a), _p(F), _p(bsum), _p(xval), _p(xind), len(xval),
                                         F.shape[1], l2, w, int(weighted))
        return g

    def calc_fun_and_grad(self, a, F, bsum, xval, xind, l2, w):
        g = np.empty(F.shape[1], self.np_t)
        f = self.lib.oracle_calc_fun_and_grad(_p(g), _p(a), _p(F), _p(bsum), _p(xval), _p(xind),
                                              len(xval), F.shape[1], l2, w)
        return f, g

    # ---- G2 single-row solvers -----------------------------------------------------------
    def cg_row(self, x, F, bsum, xval, xind, l2, w, maxupd, limit_step):
        x = x.copy()
        f = self.c_t()
        ni, nf = C.c_size_t(), C.c_size_t()
        rc = self.lib.oracle_cg_row(_p(x), _p(F), _p(bsum), _p(xval), _p(xind), len(xval), F.shape[1],
                                    l2, w, maxupd, int(limit_step), C.byref(f), C.byref(ni), C.byref(nf))
        return x, f.value, ni.value, nf.value, rc

    def tnc_row(self, x, F, bsum, xval, xind, l2, w, maxupd):
        x = x.copy()
        f = self.c_t()
        nf, ni = C.c_int(), C.c_int()
        rc = self.lib.oracle_tnc_row(_p(x), _p(F), _p(bsum), _p(xval), _p(xind), len(xval), F.shape[1],
                                     l2, w, maxupd, C.byref(f), C.byref(nf), C.byref(ni))
        return x, f.value, nf.value, ni.value, rc

    # ---- G3 half sweeps ------------------------------------------------------------------
    def sum_by_cols(self, M):
        out = np.empty(M.shape[1], self.np_t)
        self.lib.oracle_sum_by_cols(_p(out), _p(M), M.shape[0], M.shape[1])
        return out

    def pg_iteration(self, M, F, xval, indptr, indices, cnst_div, cnst_sum, bsum_w, step, w, maxupd,
                     nthreads=1):
        self.lib.oracle_pg_iteration(_p(M), _p(F), _p(xval), _p(indptr), _p(indices), M.shape[0],
                                     M.shape[1], cnst_div, _p(cnst_sum), _p(bsum_w), step, w, maxupd,
                                     nthreads)

    def cg_iteration(self, M, F, xval, indptr, indices, limit_step, bsum, l2, w, maxupd, bsum_w=None,
                     nthreads=1):
        self.lib.oracle_cg_iteration(_p(M), _p(F), _p(xval), _p(indptr), _p(indices), M.shape[0],
                                     M.shape[1], int(limit_step), _p(bsum), l2, w, maxupd, _p(bsum_w),
                                     nthreads)

    def tncg_iteration(self, M, F, reuse_prev, xval, indptr, indices, bsum, l2, w, maxupd, early_stop,
                       bsum_w=None, nthreads=1):
        return self.lib.oracle_tncg_iteration(_p(M), _p(F), int(reuse_prev), _p(xval), _p(indptr),
                                              _p(indices), M.shape[0], M.shape[1], _p(bsum), l2, w,
                                              maxupd, int(early_stop), _p(bsum_w), nthreads)

    # ---- G4 full run ---------------------------------------------------------------------
    def run_poismf(self, A, Xr, Xr_indptr, Xr_indices, B, Xc, Xc_indptr, Xc_indices, l2_reg, l1_reg,
                   w_mult, step_size, method, limit_step, numiter, maxupd, early_stop, reuse_prev,
                   handle_interrupt=True, nthreads=1):
        return self.lib.oracle_run_poismf(
            _p(A), _p(Xr), _p(Xr_indptr), _p(Xr_indices), _p(B), _p(Xc), _p(Xc_indptr), _p(Xc_indices),
            A.shape[0], B.shape[0], A.shape[1], l2_reg, l1_reg, w_mult, step_size, METHOD[method],
            bool(limit_step), numiter, maxupd, bool(early_stop), bool(reuse_prev),
            bool(handle_interrupt), nthreads)


    _fm_symbol = "oracle_factors_multiple"

    def predict_multiple(self, A, B, ixA, ixB, nthreads=1):
        out = np.empty(len(ixA), self.np_t)
        self.lib.oracle_predict_multiple(_p(out), _p(A), _p(B), _p(ixA), _p(ixB), len(ixA), A.shape[1], nthreads)
        return out

    def topn(self, a_vec, B, include_ix, exclude_ix, n_top):
        ix = np.empty(n_top, np.uint64)
        sc = np.empty(n_top, self.np_t)
        rc = self.lib.oracle_topn(_p(a_vec), _p(B), B.shape[1], _p(include_ix) if len(include_ix) else None, len(include_ix),
                                  _p(exclude_ix) if len(exclude_ix) else None, len(exclude_ix), _p(ix), _p(sc), n_top, B.shape[0])
        return rc, ix, sc

    def factors_multiple(self, B, Bsum, Amean, Xr, Xr_indptr, Xr_indices, l2_reg, w_mult, step_size, niter, maxupd,
                         method, limit_step, reuse_mean, nthreads=1):
        """ref: src/pred.c:66-199; argument meaning of poismf_c_wrapper.pxi:147-160.  Returns A [n_new x k]."""
        dimA, k = len(Xr_indptr) - 1, B.shape[1]
        A = np.full((dimA, k), np.nan, self.np_t)  # the reference hands an uninitialised array
        fn = getattr(self.lib, self._fm_symbol)
        rc = fn(_p(A), _p(B), _p(Bsum), _p(Amean), _p(Xr), _p(Xr_indptr), _p(Xr_indices), k, dimA, l2_reg, w_mult,
                step_size, niter, maxupd, METHOD[method], bool(limit_step), bool(reuse_mean), nthreads)
        assert rc == 0
        return A


class _FData(C.Structure):
    pass


def _fdata_type(c_real):
    class FData(C.Structure):  # ref: src/poismf.h:121-130
        _fields_ = [("B", C.c_void_p), ("Bsum", C.c_void_p), ("Xr", C.c_void_p), ("X_ind", C.c_void_p),
                    ("nnz_this", C.c_size_t), ("l2_reg", c_real), ("w_mult", c_real), ("k", C.c_int)]
    return FData


def ref_available(is_float=False):
    return os.path.exists(os.path.join(HERE, "_ref", "libpoismf_ref_f.so" if is_float else "libpoismf_ref_d.so"))


class Reference(_Lib):
    """The real reference, compiled in place from /root/reference/src by `make -C oracle ref`."""

    def __init__(self, is_float=False):
        path = os.path.join(HERE, "_ref", "libpoismf_ref_f.so" if is_float else "libpoismf_ref_d.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle ref` where /root/reference exists)")
        super().__init__(path, is_float)
        L, r = self.lib, self.c_t
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        self.FData = _fdata_type(r)
        L.calc_grad_pgd.argtypes = [vp, vp, vp, vp, vp, sz, i]
        L.calc_grad_pgd.restype = None
        for name in ("calc_fun_single", "calc_grad_single", "calc_grad_single_w"):
            getattr(L, name).argtypes = [vp, i, vp, vp]
            getattr(L, name).restype = None
        L.calc_fun_and_grad.argtypes = [vp, vp, vp, vp]
        L.calc_fun_and_grad.restype = i
        L.minimize_nonneg_cg.argtypes = [vp, i, vp, vp, vp, vp, vp, r, sz, sz, vp, vp, r, r, sz, C.c_bool,
                                         vp, i, i]
        L.minimize_nonneg_cg.restype = i
        L.tnc.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, r, r, r, r, r, r, r, r, vp, vp,
                          vp, vp]
        L.tnc.restype = i
        L.sum_by_cols.argtypes = [vp, vp, sz, sz]
        L.sum_by_cols.restype = None
        L.pg_iteration.argtypes = [vp, vp, vp, vp, vp, sz, sz, r, vp, vp, r, r, sz, vp, i]
        L.pg_iteration.restype = None
        L.cg_iteration.argtypes = [vp, vp, vp, vp, vp, sz, sz, C.c_bool, vp, r, r, sz, vp, vp, i]
        L.cg_iteration.restype = None
        L.tncg_iteration.argtypes = [vp, vp, C.c_bool, vp, vp, vp, sz, sz, vp, r, r, i, vp, vp, vp, vp,
                                     vp, vp, vp, i]
        L.tncg_iteration.restype = None
        L.run_poismf.argtypes = [vp] * 8 + [sz] * 3 + [r] * 4 + [i, C.c_bool, sz, sz] + [C.c_bool] * 3 + [i]
        L.run_poismf.restype = i
        L.factors_multiple.argtypes = [vp] * 7 + [i, sz, r, r, r, sz, sz, i, C.c_bool, C.c_bool, i]
        L.factors_multiple.restype = i
        L.predict_multiple.argtypes = [vp, vp, vp, vp, vp, sz, i, i]
        L.predict_multiple.restype = None
        L.topN.argtypes = [vp, vp, i, vp, sz, vp, sz, vp, vp, sz, sz, i]
        L.topN.restype = i

    def _fd(self, F, bsum, xval, xind, l2, w):
        return self.FData(F.ctypes.data, bsum.ctypes.data, xval.ctypes.data, xind.ctypes.data, len(xval),
                          l2, w, F.shape[1])

    def calc_grad_pgd(self, a, F, xval, xind):
        k = F.shape[1]
        out = np.empty(k, self.np_t)
        self.lib.calc_grad_pgd(_p(out), _p(a), _p(F), _p(xval), _p(xind), len(xval), k)
        return out

    def calc_fun_single(self, a, F, bsum, xval, xind, l2, w):
        fd = self._fd(F, bsum, xval, xind, l2, w)
        f = self.c_t()
        self.lib.calc_fun_single(_p(a), F.shape[1], C.addressof(f), C.addressof(fd))
        return f.value

    def calc_grad_single(self, a, F, bsum, xval, xind, l2, w, weighted=False):
        fd = self._fd(F, bsum, xval, xind, l2, w)
        g = np.empty(F.shape[1], self.np_t)
        fn = self.lib.calc_grad_single_w if weighted else self.lib.calc_grad_single
        fn(_p(a), F.shape[1], _p(g), C.addressof(fd))
        return g

    def calc_fun_and_grad(self, a, F, bsum, xval, xind, l2, w):
        fd = self._fd(F, bsum, xval, xind, l2, w)
        g = np.empty(F.shape[1], self.np_t)
        f = self.c_t()
        self.lib.calc_fun_and_grad(_p(a), C.addressof(f), _p(g), C.addressof(fd))
        return f.value, g

    def cg_row(self, x, F, bsum, xval, xind, l2, w, maxupd, limit_step):
        """minimize_nonneg_cg with cg_iteration's constants (ref: src/poismf.c:315-320)."""
        x = x.copy()
        k = F.shape[1]
        fd = self._fd(F, bsum, xval, xind, l2, w)
        f = self.c_t()
        ni, nf = C.c_size_t(), C.c_size_t()
        buf = np.empty(5 * k, self.np_t)
        gfun = self.lib.calc_grad_single_w if w != 1.0 else self.lib.calc_grad_single
        rc = self.lib.minimize_nonneg_cg(
            _p(x), k, C.addressof(f), C.cast(self.lib.calc_fun_single, C.c_void_p),
            C.cast(gfun, C.c_void_p), None, C.addressof(fd), 1e-2, 150, maxupd, C.addressof(ni),
            C.addressof(nf), 0.25, 0.01, 20, bool(limit_step), _p(buf), 1, 0)
        return x, f.value, ni.value, nf.value, rc

    def tnc_row(self, x, F, bsum, xval, xind, l2, w, maxupd):
        """tnc with tncg_iteration's constants (ref: src/poismf.c:383-391)."""
        x = x.copy()
        k = F.shape[1]
        fd = self._fd(F, bsum, xval, xind, l2, w)
        f = self.c_t()
        nf, ni = C.c_int(), C.c_int()
        buf = np.empty(22 * k, self.np_t)
        ibuf = np.empty(k, np.int32)
        zeros = np.zeros(k, self.np_t)
        infs = np.full(k, np.inf, self.np_t)
        max_cg = int(max(1.0, min(50.0, k / 2.0)))
        rc = self.lib.tnc(k, _p(x), C.addressof(f), buf.ctypes.data + 21 * k * buf.itemsize,
                          C.cast(self.lib.calc_fun_and_grad, C.c_void_p), C.addressof(fd), _p(zeros),
                          _p(infs), None, None, 0, max_cg, maxupd, 0.25, 10.0, 0.0, 0.0, 1e-4, -1.0, -1.0,
                          1.3, C.addressof(nf), C.addressof(ni), _p(buf), _p(ibuf))
        return x, f.value, nf.value, ni.value, rc

    def sum_by_cols(self, M):
        out = np.empty(M.shape[1], self.np_t)
        self.lib.sum_by_cols(_p(out), _p(M), M.shape[0], M.shape[1])
        return out

    def pg_iteration(self, M, F, xval, indptr, indices, cnst_div, cnst_sum, bsum_w, step, w, maxupd,
                     nthreads=1):
        k = M.shape[1]
        buf = np.empty(k * nthreads, self.np_t)
        self.lib.pg_iteration(_p(M), _p(F), _p(xval), _p(indptr), _p(indices), M.shape[0], k, cnst_div,
                              _p(cnst_sum), _p(bsum_w), step, w, maxupd, _p(buf), nthreads)

    def cg_iteration(self, M, F, xval, indptr, indices, limit_step, bsum, l2, w, maxupd, bsum_w=None,
                     nthreads=1):
        k = M.shape[1]
        buf = np.empty(5 * k * nthreads, self.np_t)
        self.lib.cg_iteration(_p(M), _p(F), _p(xval), _p(indptr), _p(indices), M.shape[0], k,
                              bool(limit_step), _p(bsum), l2, w, maxupd, _p(buf), _p(bsum_w), nthreads)

    def tncg_iteration(self, M, F, reuse_prev, xval, indptr, indices, bsum, l2, w, maxupd, early_stop,
                       bsum_w=None, nthreads=1):
        k = M.shape[1]
        buf = np.empty(22 * k * nthreads, self.np_t)
        ibuf = np.empty(k * nthreads, np.int32)
        unch = np.empty(k * nthreads, self.np_t) if early_stop else None
        conv = C.c_bool(False)
        zeros = np.zeros(k, self.np_t)
        infs = np.full(k, np.inf, self.np_t)
        self.lib.tncg_iteration(_p(M), _p(F), bool(reuse_prev), _p(xval), _p(indptr), _p(indices),
                                M.shape[0], k, _p(bsum), l2, w, maxupd, _p(buf), _p(ibuf), _p(unch),
                                C.addressof(conv), _p(zeros), _p(infs), _p(bsum_w), nthreads)
        return int(conv.value)

    def run_poismf(self, A, Xr, Xr_indptr, Xr_indices, B, Xc, Xc_indptr, Xc_indices, l2_reg, l1_reg,
                   w_mult, step_size, method, limit_step, numiter, maxupd, early_stop, reuse_prev,
                   handle_interrupt=True, nthreads=1):
        return self.lib.run_poismf(
            _p(A), _p(Xr), _p(Xr_indptr), _p(Xr_indices), _p(B), _p(Xc), _p(Xc_indptr), _p(Xc_indices),
            A.shape[0], B.shape[0], A.shape[1], l2_reg, l1_reg, w_mult, step_size, METHOD[method],
            bool(limit_step), numiter, maxupd, bool(early_stop), bool(reuse_prev),
            bool(handle_interrupt), nthreads)

    _fm_symbol = "factors_multiple"

    def factors_multiple(self, B, Bsum, Amean, Xr, Xr_indptr, Xr_indices, l2_reg, w_mult, step_size, niter, maxupd,
                         method, limit_step, reuse_mean, nthreads=1):
        """ref: src/pred.c:66-199; argument meaning of poismf_c_wrapper.pxi:147-160.  Returns A [n_new x k]."""
        dimA, k = len(Xr_indptr) - 1, B.shape[1]
        A = np.full((dimA, k), np.nan, self.np_t)  # the reference hands an uninitialised array
        fn = getattr(self.lib, self._fm_symbol)
        rc = fn(_p(A), _p(B), _p(Bsum), _p(Amean), _p(Xr), _p(Xr_indptr), _p(Xr_indices), k, dimA, l2_reg, w_mult,
                step_size, niter, maxupd, METHOD[method], bool(limit_step), bool(reuse_mean), nthreads)
        assert rc == 0
        return A

    def predict_multiple(self, A, B, ixA, ixB, nthreads=1):
        out = np.empty(len(ixA), self.np_t)
        self.lib.predict_multiple(_p(out), _p(A), _p(B), _p(ixA), _p(ixB), len(ixA), A.shape[1], nthreads)
        return out

    def topn(self, a_vec, B, include_ix, exclude_ix, n_top):
        ix = np.empty(n_top, np.uint64)
        sc = np.empty(n_top, self.np_t)
        inc = include_ix.copy()
        exc = exclude_ix.copy()  # the reference sorts / permutes its index arguments in place
        rc = self.lib.topN(_p(a_vec), _p(B), B.shape[1], _p(inc) if len(inc) else None, len(inc), _p(exc) if len(exc) else None,
                           len(exc), _p(ix), _p(sc), n_top, B.shape[0], 1)
        return rc, ix, sc
