"""The boundary's host <-> device copies (poismf_amd/csrc/devmem.hpp): arrays of 16 MB and more travel through pinned chunks
filled by a few host threads -- indices narrowed from size_t to u32 on the way -- and the factors come back the same way.  The
copies must be invisible: run_poismf on the same input gives the same BITS whether the staged path is taken (forced here by
lowering its threshold so that every array is cut into many chunks, with ragged last chunks per thread), not taken at all, or
taken with another number of threads.  The knobs are read once per process, hence the children.  Needs an MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from tests import helpers as H
from tests.test_gpu_parity import gpu_run
dimA, dimB, k = 30000, 9000, 50
csr, csc, A0, B0 = H.small_problem(dimA, dimB, 1500000, k, {prec!r}, seed=21, powerlaw=True, empty_rows=(5, 29999))
A, B, _ = gpu_run(csr, csc, A0, B0, {method!r}, 2, k, **{kw!r})
np.save({out!r}, np.concatenate([A.ravel(), B.ravel()]))
"""


# (PG with the step / l2 of bench.py's "finite" block: the Python defaults drive every factor entry to exact zero, which would
# compare equal whatever the copies did)
@pytest.mark.parametrize("prec,method,kw", [(True, "pg", dict(l2_reg=1e3, step_size=1e-9)), (False, "cg", {})], ids=["f32-pg", "f64-cg"])
def test_staged_copies_leave_no_trace(prec, method, kw, tmp_path):
    runs = {
        "plain": {"POISMF_HIP_NO_STAGED_UPLOAD": "1"},
        # 1.5M nonzeros: 6 MB of indices / 6-12 MB of values per half, 6-12 MB of factors; 4 MB chunks, so 3 and 7 threads
        # give every thread a ragged single chunk or a few chunks with a short tail
        "staged3": {"POISMF_HIP_STAGED_MIN_BYTES": "1", "POISMF_HIP_HOST_THREADS": "3"},
        "staged7": {"POISMF_HIP_STAGED_MIN_BYTES": "1", "POISMF_HIP_HOST_THREADS": "7"},
        "staged1": {"POISMF_HIP_STAGED_MIN_BYTES": "1", "POISMF_HIP_HOST_THREADS": "1"},
        # round 4: run_poismf uploads the A side's matrix on the second stream while the first B half runs; here the whole matrix first
        "no_overlap": {"POISMF_HIP_NO_UPLOAD_OVERLAP": "1"},
        "no_overlap_staged": {"POISMF_HIP_NO_UPLOAD_OVERLAP": "1", "POISMF_HIP_STAGED_MIN_BYTES": "1", "POISMF_HIP_HOST_THREADS": "3"},
    }
    res = {}
    for tag, env in runs.items():
        out = str(tmp_path / f"{tag}.npy")
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, prec=prec, method=method, kw=kw, out=out)], check=True, env=e, cwd=ROOT,
                       timeout=600)
        res[tag] = np.load(out)
    assert np.isfinite(res["plain"]).all() and res["plain"].any()
    for tag in ("staged3", "staged7", "staged1", "no_overlap", "no_overlap_staged"):
        assert np.array_equal(res[tag], res["plain"]), tag


CHILD_CACHE = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from tests import helpers as H
from tests.test_gpu_parity import gpu_run
from poismf_amd import api
outs = []
# the same shape three times (the second and third calls find every large array in the cache, with the previous call's contents in
# it), another shape in between (its arrays join the list), a third precision-independent shape after an explicit release
for dimA, dimB, nnz, seed in ((30000, 9000, 1500000, 21), (30000, 9000, 1500000, 22), (12000, 20000, 900000, 23), (30000, 9000, 1500000, 21)):
    csr, csc, A0, B0 = H.small_problem(dimA, dimB, nnz, 50, False, seed=seed, powerlaw=True)
    A, B, _ = gpu_run(csr, csc, A0, B0, "cg", 2, 50)
    outs.append(np.concatenate([A.ravel(), B.ravel()]))
    if seed == 23:
        api.load_library(False).poismf_hip_release_cache()
np.save({out!r}, np.concatenate(outs))
"""


def test_kept_device_arrays_leave_no_trace(tmp_path):
    """A caller that opts in (POISMF_HIP_DEVICE_CACHE_MB; off by default since round 5) has devmem.hpp keep the device arrays of finished
    calls for the next call of the same sizes: same bits as with the list off, and the same problem gives the same bits the first and
    the last time."""
    res = {}
    for tag, env in {"kept": {"POISMF_HIP_DEVICE_CACHE_MB": "16384"}, "off": {}, "tiny": {"POISMF_HIP_DEVICE_CACHE_MB": "40"}}.items():
        out = str(tmp_path / f"{tag}.npy")
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", CHILD_CACHE.format(root=ROOT, out=out)], check=True, env=e, cwd=ROOT, timeout=600)
        res[tag] = np.load(out)
    assert np.isfinite(res["off"]).all() and res["off"].any()
    assert np.array_equal(res["kept"], res["off"]) and np.array_equal(res["tiny"], res["off"])
    n = (30000 + 9000) * 50
    assert np.array_equal(res["kept"][:n], res["kept"][-n:])           # first and last call: the same problem
    assert not np.array_equal(res["kept"][:n], res["kept"][n:2 * n])   # (and the second one is another)


CHILD_HANDOUT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from tests import helpers as H
from tests.test_gpu_parity import gpu_run
# 300 item rows of ~800 nonzeros (PG fp32, k = 50: the four-wave lane launch) next to 4000 user rows of ~60 (one-wave register kernels)
csr, csc, A0, B0 = H.small_problem(4000, 300, 240000, 50, True, seed=31)
A, B, _ = gpu_run(csr, csc, A0, B0, "pg", 3, 50, l2_reg=1e3, step_size=1e-9, maxupd=10)
np.save({out!r}, np.concatenate([A.ravel(), B.ravel()]))
"""


def test_pg_row_hand_out_leaves_no_trace(tmp_path):
    """PG's multi-wave lane launches hand their rows out one per workgroup (the hardware dispatcher decides who gets which, round 4; the
    persistent-workgroup variants went in round 6): two fresh processes, and one that keeps every launch on one stream, give the same bits --
    a row's arithmetic depends on its length class alone."""
    res = {}
    for tag, env in {"fresh": {}, "again": {}, "one_stream": {"POISMF_HIP_NO_FORK": "1"}}.items():
        out = str(tmp_path / f"{tag}.npy")
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", CHILD_HANDOUT.format(root=ROOT, out=out)], check=True, env=e, cwd=ROOT, timeout=600)
        res[tag] = np.load(out)
    assert np.isfinite(res["fresh"]).all() and res["fresh"].any()
    for tag in ("again", "one_stream"):
        assert np.array_equal(res[tag], res["fresh"]), tag


CHILD_LIMIT = r"""
import sys, numpy as np, ctypes
sys.path.insert(0, {root!r})
from tests import helpers as H
from tests.test_gpu_parity import gpu_run
from poismf_amd import api
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value >> 20
csr, csc, A0, B0 = H.small_problem(60000, 20000, 6000000, 50, False, seed=5)
outs, free = [], []
gpu_run(csr, csc, A0, B0, "pg", 1, 50)          # (device up, pinned chunks and streams allocated: what the process keeps either way)
free.append(free_mb())
A, B, _ = gpu_run(csr, csc, A0, B0, "pg", 1, 50)
outs.append(np.concatenate([A.ravel(), B.ravel()]))
free.append(free_mb())                          # default: nothing of the call is left on the device
assert api.set_device_cache_mb(4096, False) == {{False: 0}}
A, B, _ = gpu_run(csr, csc, A0, B0, "pg", 1, 50)
outs.append(np.concatenate([A.ravel(), B.ravel()]))
free.append(free_mb())                          # opted in: the call's large arrays are still allocated
A, B, _ = gpu_run(csr, csc, A0, B0, "pg", 1, 50)
outs.append(np.concatenate([A.ravel(), B.ravel()]))
api.release_cache()
free.append(free_mb())
assert api.set_device_cache_mb(0, False) == {{False: 4096}}
np.save({out!r}, np.concatenate(outs))
np.save({out!r} + ".free.npy", np.array(free))
"""


def test_nothing_survives_a_call_unless_the_caller_opts_in(tmp_path):
    """"no handles/state survive the call" (ref src/poismf.c:610-617).  By default run_poismf leaves the device's free
    memory where it found it; after poismf_hip_set_device_cache_mb() the call's arrays stay (~100 MB here) until release_cache();
    the results are the same bits either way."""
    out = str(tmp_path / "limit.npy")
    subprocess.run([sys.executable, "-c", CHILD_LIMIT.format(root=ROOT, out=out)], check=True, cwd=ROOT, timeout=600)
    res, free = np.load(out), np.load(out + ".free.npy")
    n = res.size // 3
    assert np.array_equal(res[:n], res[n:2 * n]) and np.array_equal(res[:n], res[2 * n:])
    assert abs(int(free[1]) - int(free[0])) <= 8, free            # default: free memory back where it was (MB)
    assert int(free[0]) - int(free[2]) >= 60, free                # kept: the X arrays of 6e6 nonzeros (~72 MB CSR + CSC) are still there
    assert abs(int(free[3]) - int(free[0])) <= 8, free            # released
