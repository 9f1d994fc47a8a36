"""CPU-side checks of the boundary: the two shared libraries build, load, and export every symbol that
include/poismf_hip.h declares; the host mirror of the reference's wrapper keeps its calling convention.
No compute call is made here (that needs a GPU: tests/test_gpu_parity.py)."""
import inspect
import os
import re

import numpy as np
import pytest

from poismf_amd import api, build, harness, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "poismf_hip.h")).read()
    return sorted(set(re.findall(r"^POISMF_HIP_API\s+[\w\s\*]*?\b(\w+)\s*\(", text, re.M)))


@pytest.mark.parametrize("use_float", [False, True, "r"])
def test_library_exports_header(use_float):
    """the two Python flavours and the R-ABI flavour (int indices, ref src/poismf.h:75-89)"""
    lib = api.load_library(use_float)
    declared = _declared_symbols()
    assert sorted(api.EXPORTED_SYMBOLS) == declared
    for name in declared:
        assert getattr(lib, name) is not None


def test_run_poismf_signature_matches_cython_wrapper():
    """positional order and defaults of ref poismf/poismf_c_wrapper.pxi:57-72"""
    sig = inspect.signature(api._run_poismf)
    assert list(sig.parameters) == ["Xr", "Xr_indices", "Xr_indptr", "Xc", "Xc_indices", "Xc_indptr", "A", "B",
                                    "method", "limit_step", "l2_reg", "l1_reg", "w_mult", "step_size", "niter",
                                    "maxupd", "early_stop", "reuse_prev", "handle_interrupt", "nthreads"]
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    assert d == dict(method="tncg", limit_step=0, l2_reg=1e9, l1_reg=0, w_mult=1., step_size=1e-7, niter=10,
                     maxupd=1, early_stop=1, reuse_prev=1, handle_interrupt=1, nthreads=1)


def test_wrapper_checks_before_touching_the_gpu():
    e = np.empty(0, np.float64)
    i = np.zeros(1, np.uint64)
    with pytest.raises(ValueError, match="no non-zero"):
        api._run_poismf(e, i, i, e, i, i, np.ones((1, 2)), np.ones((1, 2)))
    with pytest.raises(TypeError):
        x = np.ones(1, np.float32)  # dtype mismatch with A (float64)
        api._run_poismf(x, i, i, x, i, i, np.ones((1, 2)), np.ones((1, 2)))


def test_auto_defaults_table():
    """ref: poismf/__init__.py:250-255"""
    assert harness.auto_defaults("tncg", 50) == (1e3, 750, 10)
    assert harness.auto_defaults("cg", 50) == (1e4, 5, 30)
    assert harness.auto_defaults("pg", 50) == (1e9, 10, 10)


def test_process_data_sums_duplicates_and_sorts():
    coo = synth.readme_coo()
    csr, csc = harness.process_data(coo, True)
    assert csr[0].dtype == np.float32 and csr[1].dtype == np.uint64 and csr[2].dtype == np.uint64
    assert len(csr[0]) == len(csc[0]) == 9490 and csr[0].sum() == coo.data.sum()
    for data, idx, ptr in (csr, csc):
        for r in range(len(ptr) - 1):
            seg = idx[ptr[r]:ptr[r + 1]]
            assert np.all(seg[1:] > seg[:-1])


def test_initialize_matrices_stream():
    A, B = harness.initialize_matrices(3, 4, 2, False, 1)
    rng = np.random.default_rng(1)
    assert np.array_equal(A, 0.3 + rng.uniform(0, 0.01, (3, 2))) and np.array_equal(B, 0.3 + rng.uniform(0, 0.01, (4, 2)))


def test_no_gpu_means_loud_failure():
    """the product path has no CPU fallback: without a device the call must raise, not compute"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    coo = synth.readme_coo()
    csr, csc = harness.process_data(coo, False)
    A, B = harness.initialize_matrices(100, 1000, 5, False, 1)
    with pytest.raises(MemoryError):
        api._run_poismf(csr[0], csr[1], csr[2], csc[0], csc[1], csc[2], A, B, "pg", True, 1e9, 0., 1., 1e-7, 1, 1)
