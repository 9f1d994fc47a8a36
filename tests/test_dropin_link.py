"""The reference-side binding of INTEGRATION.md §1 is built for real (build container only: it needs /root/reference, which never
travels): scripts/verify_dropin_link.sh copies the reference to a temp dir, builds its Cython wrapper against libpoismf_hip_{d,f}.so
with scripts/dropin/setup_hip.py, imports both modules with LD_BIND_NOW=1 and reads LD_DEBUG=bindings to see that the wrapper's
run_poismf / factors_multiple / predict_multiple / topN (ref poismf_c_wrapper.pxi:95-103, :192-200, :112, :241) land in the HIP
library while factors_single keeps the CPU core (ref src/pred.c:281-284)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "poismf.c")), reason="the reference is only present in the build container")
def test_reference_wrapper_links_against_the_hip_libraries():
    pytest.importorskip("Cython")
    from poismf_amd import build
    build.build()
    res = subprocess.run([os.path.join(ROOT, "scripts", "verify_dropin_link.sh"), REF], capture_output=True, text=True, timeout=600)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out
    for mod, lib in (("c_funs_double", "libpoismf_hip_d.so"), ("c_funs_float", "libpoismf_hip_f.so")):
        for sym in ("run_poismf", "factors_multiple", "predict_multiple", "topN"):
            assert f"bound: {mod}.{sym} -> {lib}" in out, out
    assert "factors_single on the CPU core, float64" in out and "factors_single on the CPU core, float32" in out
    assert "verify_dropin_link: OK" in out


def test_integration_md_quotes_the_recipe_the_script_runs():
    """INTEGRATION.md's recipe IS scripts/dropin/setup_hip.py: the renames and the source list quoted there are read from it"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    setup = open(os.path.join(ROOT, "scripts", "dropin", "setup_hip.py")).read()
    assert "scripts/dropin/setup_hip.py" in text and "scripts/verify_dropin_link.sh" in text
    for name in ("run_poismf", "factors_multiple", "predict_multiple", "topN"):
        assert f"-D{name}={name}_cpu" in text
        assert f'"{name}"' in setup
    for src in ("src/poismf.c", "src/nonnegcg.c", "src/tnc.c", "src/pred.c", "src/topN.c"):
        assert src in setup
