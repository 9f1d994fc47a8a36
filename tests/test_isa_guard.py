"""The ISA guard of the team kernels (scripts/isa_guard.py, run by poismf_amd.build.build() on every library it links): on the in-tree
libraries it must hold, on synthetic listings that break either rule it must fail.  CPU only: llvm-objdump reads the code objects."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _guard():
    spec = importlib.util.spec_from_file_location("pmf_isa_guard", os.path.join(ROOT, "scripts", "isa_guard.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


KERNEL = "void half_sweep_lane_team_kernel<double, 1, 50, 1, 0, 0, 4, 32>(HalfArgs<double>)"
CALLEE = "pmf::TeamVals<2> pmf::team_sum_call<2>(unsigned long long*, int)"


def _listing(funcs):
    out = []
    for i, (name, insns) in enumerate(funcs.items()):
        out.append(f"{0x1000 * (i + 1):016x} <{name}>:")
        out += [f"\t{ins}   // {0x1000 * (i + 1) + 4 * j:012X}: DEADBEEF" for j, ins in enumerate(insns)]
        out.append("")
    return "\n".join(out)


def test_guard_accepts_a_called_exchange_without_narrowed_spills():
    g = _guard()
    good = _listing({KERNEL: ["s_load_dwordx2 s[0:1], s[4:5], 0x0", "scratch_store_dword off, v0, off offset:4", "s_swappc_b64 s[30:31], s[2:3]", "s_endpgm"],
                     CALLEE: ["scratch_store_dword off, v40, off", "s_and_saveexec_b64 s[2:3], s[4:5]", "v_mov_b32 v0, v1", "s_or_b64 exec, exec, s[2:3]",
                              "scratch_load_dword v40, off, off", "s_setpc_b64 s[30:31]"]})
    msgs = []
    assert g.check_listing(good, "synthetic", msgs.append), msgs


def test_guard_rejects_an_inlined_exchange_and_a_spill_under_a_narrowed_mask():
    g = _guard()
    inlined = _listing({KERNEL: ["s_load_dwordx2 s[0:1], s[4:5], 0x0", "s_endpgm"]})
    msgs = []
    assert not g.check_listing(inlined, "synthetic", msgs.append)
    assert any("inlined" in m for m in msgs)
    spilled = _listing({KERNEL: ["s_swappc_b64 s[30:31], s[2:3]", "s_endpgm"],
                        CALLEE: ["s_and_saveexec_b64 s[2:3], s[4:5]", "scratch_store_dword off, v40, off offset:8", "s_or_b64 exec, exec, s[2:3]", "s_setpc_b64 s[30:31]"]})
    msgs = []
    assert not g.check_listing(spilled, "synthetic", msgs.append)
    assert any("EXEC may be narrowed" in m for m in msgs)
    # a loop's mask bookkeeping (s_andn2 on EXEC) counts as narrowed until the saved mask is back
    looped = _listing({KERNEL: ["s_swappc_b64 s[30:31], s[2:3]"],
                       CALLEE: ["s_mov_b64 s[6:7], exec", "s_andn2_b64 exec, exec, s[8:9]", "scratch_store_dword off, v1, off", "s_mov_b64 exec, s[6:7]"]})
    assert not g.check_listing(looped, "synthetic", [].append)


def test_guard_holds_on_the_in_tree_libraries():
    g = _guard()
    if not os.path.exists(g.OBJDUMP):
        pytest.skip("no llvm-objdump here")
    from poismf_amd import build
    build.build()
    lib = build.lib_path(False)
    msgs = []
    assert g.check_library(lib, msgs.append), [m for m in msgs if ": info: " not in m]
    # it looked at the real thing: the double flavour has the lane-team kernel and its out-of-line exchange
    assert any("half_sweep_lane_team_kernel" in m and "calls" in m for m in msgs), msgs


def test_build_runs_the_guard():
    src = open(os.path.join(ROOT, "poismf_amd", "build.py")).read()
    assert "isa_guard(out)" in src and "def isa_guard(lib)" in src
