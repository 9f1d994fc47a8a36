"""Several devices behind run_poismf() itself (poismf_hip_host.hip, run_poismf_multi): POISMF_HIP_DEVICES lists them, the rows of A
and of B are cut into one nnz-balanced range per entry, every entry gets a session with its shard and replicas of both factors,
and after each half the updated rows travel device to device.  One GPU is what a test box has, so the list names it twice or
three times: every line of the path runs (shards, host threads, peer copies, events, re-padding, the summed early-stop counter);
what cannot be shown here is only that the copies cross xGMI.  Results must equal the single-session call bit for bit.
Needs an MI355X."""
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
csr, csc, A0, B0 = H.small_problem(1500, 900, 60000, {k}, {prec}, seed=8, powerlaw=True, empty_rows=(5,))
A, B, _ = gpu_run(csr, csc, A0, B0, {method!r}, 3, {k}, **{kw!r})
np.save({out!r}, np.concatenate([A.ravel().astype(np.float64), B.ravel().astype(np.float64)]))
"""


@pytest.mark.parametrize("method,prec,k,kw", [("pg", True, 50, {}), ("cg", False, 50, {}), ("cg", True, 20, {}),
                                              ("tncg", False, 20, dict(maxupd=60, early_stop=True))])
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_run_poismf_over_a_device_list_equals_the_single_device_call(tmp_path, method, prec, k, kw, devices):
    res = {}
    # (POISMF_SHARD_COLSUM_MIN_ROWS=1: the first stage of the column sums shared between the devices for factors of any size, as it is for
    # factors of 262 144 rows and more by default -- partial sums pushed peer to peer, same bits)
    for tag, env in (("one", {}), ("many", {"POISMF_HIP_DEVICES": devices, "POISMF_SHARD_COLSUM_MIN_ROWS": "1"})):
        out = str(tmp_path / f"{tag}.npy")
        e = dict(os.environ); e.pop("POISMF_HIP_DEVICES", None); e.update(env)
        subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, out=out, method=method, prec=prec, k=k, kw=kw)], check=True, env=e,
                       cwd=ROOT, timeout=600)
        res[tag] = np.load(out)
    assert np.isfinite(res["one"]).all()
    assert np.array_equal(res["one"], res["many"])


CHILD8 = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from tests import helpers as H
from tests.test_gpu_parity import gpu_run
# config-C5-shaped: power-law item degrees (the first items hold most nonzeros: the nnz-balanced B ranges are a few dozen rows for the first
# devices and thousands for the last), short user rows
csr, csc, A0, B0 = H.small_problem(6000, 2500, 240000, {k}, {prec}, seed=28, powerlaw=True, empty_rows=(7, 4000))
A, B, _ = gpu_run(csr, csc, A0, B0, {method!r}, 2, {k}, **{kw!r})
np.save({out!r}, np.concatenate([A.ravel().astype(np.float64), B.ravel().astype(np.float64)]))
"""


@pytest.mark.parametrize("method,prec,k,kw", [("pg", True, 50, {}), ("tncg", False, 100, dict(maxupd=80, early_stop=True)), ("cg", False, 50, {})])
def test_eight_entry_device_list_on_power_law_rows(tmp_path, method, prec, k, kw):
    """The node the north-star names has eight GPUs: POISMF_HIP_DEVICES with EIGHT entries (all naming the one GPU a test box has) on a
    power-law matrix -- eight persistent host threads, eight sessions, 56 peer copies per half, very unequal row ranges (nnz-balanced),
    the A half in four segments.  Bit for bit the single-device result.  (Unmeasured on hardware: no multi-GPU box was available to
    this build; what one device cannot show is that the copies cross xGMI.)"""
    res = {}
    for tag, env in (("one", {}), ("eight", {"POISMF_HIP_DEVICES": "0,0,0,0,0,0,0,0", "POISMF_SHARD_COLSUM_MIN_ROWS": "1" if method != "cg" else "1000000000"})):
        out = str(tmp_path / f"{tag}.npy")
        e = dict(os.environ); e.pop("POISMF_HIP_DEVICES", None); e.update(env)
        subprocess.run([sys.executable, "-c", CHILD8.format(root=ROOT, out=out, method=method, prec=prec, k=k, kw=kw)], check=True, env=e,
                       cwd=ROOT, timeout=900)
        res[tag] = np.load(out)
    assert np.isfinite(res["one"]).all()
    assert np.array_equal(res["one"], res["eight"])


def test_declared_partials_are_used_for_their_own_factor_only():
    """poismf_hip_session_partials_ready() is a one-shot promise about ONE factor's column sums (the last poismf_hip_session_colsum_partial's).
    Round 5 kept a bare flag: a half-sweep over the OTHER factor, or one after that factor had been rewritten, took foreign / stale partial sums
    for its own (advisor, round 5).  Now the promise names its factor: the half it belongs to consumes it (same bits as the full sum), any
    other half computes its own first stage, and set_factors / factors_dirty withdraw it."""
    from poismf_amd import api
    from tests import helpers as H
    csr, csc, A0, B0 = H.small_problem(5000, 3000, 200000, 50, False, seed=9)
    s = api.Session(csr, csc, A0.shape[0], B0.shape[0], 50, False)
    prm = s.make_params("cg", 1e4, maxupd=3)

    def half(which, prepare=None):
        s.set_factors(A0, B0)
        if prepare is not None:
            prepare()
        s.half_sweep(which, prm, 1e-7, 1.0)
        return s.get_factors()

    ref = {w: half(w) for w in (0, 1)}

    def declare(which):
        def go():
            s.colsum_partial(which, 0, s.colsum_blocks(which))   # every block, computed here: the honest case
            s.partials_ready()
        return go

    for w in (0, 1):     # the promise kept: same bits as the sum the half computes itself
        A, B = half(w, declare(w))
        assert np.array_equal(A, ref[w][0]) and np.array_equal(B, ref[w][1]), w
    for w in (0, 1):     # the promise made for the OTHER half's factor: declined, the half sums its own factor
        A, B = half(w, declare(1 - w))
        assert np.array_equal(A, ref[w][0]) and np.array_equal(B, ref[w][1]), w

    def stale():         # partials of A declared, then A rewritten through set_factors: withdrawn
        s.colsum_partial(0, 0, s.colsum_blocks(0))
        s.partials_ready()
        s.set_factors(A0 * 2.0, B0)
    A, B = half(0, stale)
    s.set_factors(A0 * 2.0, B0)
    s.half_sweep(0, prm, 1e-7, 1.0)
    A2, B2 = s.get_factors()
    assert np.array_equal(B, B2)
    s.close()
