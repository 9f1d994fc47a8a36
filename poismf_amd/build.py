"""Builds the HIP shared libraries in-tree (poismf_amd/libpoismf_hip_{d,f}.so) with hipcc for gfx950.

    python -m poismf_amd.build [--force] [-v]

Six translation units per precision -- the host side (poismf_hip_host.hip), one per inner solver (the row kernels of
PG, CG and TNCG are the bulk of the compile time; poismf_hip.hip is compiled once for each with -DPMF_TU=...), the
rocPRIM-based COO conversion and the serving kernels -- are compiled to object files side by side and linked.  Every object and
library carries a `.stamp` with the digest of its command line and sources: an object is rebuilt exactly when that
digest changes (a different POISMF_HIP_EXTRA_FLAGS rebuilds everything it reaches; file times play no part).  A full
build takes ~2 minutes on 8 cores.
"""
import fcntl
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(os.path.dirname(HERE), "include", "poismf_hip.h")
_ROW = ["poismf_hip.hip", "plan.hpp", "devmem.hpp", "solvers.hpp", "row_eval.hpp", "reg_eval.hpp", "lane_eval.hpp", "wave_ops.hpp"]
_HOST = ["poismf_hip_host.hip", "plan.hpp", "devmem.hpp", "row_eval.hpp", "wave_ops.hpp"]
# unit -> (source files, first is the one compiled; extra flags).  poismf_hip.hip is compiled four times: one
# translation unit per inner solver (its row kernels are the bulk of the compile time); the host side is its own file.
UNITS = {
    "poismf_hip_host": (_HOST, []),
    "poismf_hip_tncg": (_ROW, ["-DPMF_TU=1"]),
    "poismf_hip_cg": (_ROW, ["-DPMF_TU=2"]),
    "poismf_hip_pg": (_ROW, ["-DPMF_TU=3"]),
    "poismf_hip_eval": (_ROW, ["-DPMF_TU=4"]),   # evaluation-only kernels behind poismf_hip_debug_row_eval (testing aid)
    "coo_convert": (["coo_convert.hip", "devmem.hpp"], []),
    "serve": (["serve.hip", "devmem.hpp"], []),
}


# flavour -> flags: the reference builds its core twice for Python (ref: setup.py:225-243) and once for R with int indices
# (ref: src/poismf.h:75-89); the R flavour differs on the host side only and links the double flavour's row kernels
FLAVOURS = {"d": [], "f": ["-DUSE_FLOAT"], "r": ["-D_FOR_R"]}
_KERNEL_UNITS = ("poismf_hip_tncg", "poismf_hip_cg", "poismf_hip_pg", "poismf_hip_all")


def _flavour(use_float):
    return use_float if isinstance(use_float, str) else ("f" if use_float else "d")


def lib_path(use_float):
    """use_float: False / True (the two Python flavours) or a flavour name "d" / "f" / "r" """
    return os.path.join(HERE, f"libpoismf_hip_{_flavour(use_float)}.so")


def _obj_path(unit, flavour):
    flavour = _flavour(flavour)
    if flavour == "r" and unit in _KERNEL_UNITS:
        flavour = "d"
    return os.path.join(CSRC, f"{unit}_{flavour}.o")


def _digest(parts, files):
    """sha256 over strings and file contents"""
    h = hashlib.sha256(repr(parts).encode())
    for path in files:
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0" + fh.read())
    return h.hexdigest()


def _fresh(out, digest):
    """`out` exists and was produced from exactly the inputs `digest` stands for (sources AND flags: a changed
    POISMF_HIP_EXTRA_FLAGS rebuilds, file times play no part)"""
    try:
        return os.path.exists(out) and open(out + ".stamp").read().strip() == digest
    except OSError:
        return False


def _mark(out, digest):
    with open(out + ".stamp", "w") as fh:
        fh.write(digest + "\n")


STAMP = os.path.join(HERE, ".build_stamp")   # hash of the sources + flags the in-tree libraries were built from


def _units():
    """The -DPMF_TIMING development build keeps its phase timers in one device-side array, so it stays one translation unit."""
    if "-DPMF_TIMING" in os.environ.get("POISMF_HIP_EXTRA_FLAGS", "").split():
        return {"poismf_hip_host": UNITS["poismf_hip_host"], "poismf_hip_all": (_ROW, []), "coo_convert": UNITS["coo_convert"],
                "serve": UNITS["serve"]}
    return UNITS


def _flags():
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-const-variable",
             "-Wno-pass-failed"]
    return flags + os.environ.get("POISMF_HIP_EXTRA_FLAGS", "").split()   # development: e.g. -DPMF_REG_G=8, -DPMF_TIMING


def _source_hash():
    names = sorted({f for files, _ in _units().values() for f in files})
    return _digest([_flags(), sorted((u, f) for u, (_, f) in _units().items())], [os.path.join(CSRC, f) for f in names] + [HEADER])


def up_to_date():
    """The libraries exist and were built from exactly the sources + flags that are here now.  Object files do not
    travel with a snapshot of the tree (and file times do not survive one), so this is what decides whether a process
    on another machine -- or N ranks of one job at once -- may skip the compiler."""
    try:
        return all(os.path.exists(lib_path(f)) for f in FLAVOURS) and open(STAMP).read().strip() == _source_hash()
    except OSError:
        return False


def build(force=False, verbose=False):
    if not force and up_to_date():
        return lib_path(False), lib_path(True)
    with open(os.path.join(HERE, ".build_lock"), "w") as lock:   # one builder at a time (ranks of a multi-GPU job)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and up_to_date():
            return lib_path(False), lib_path(True)
        # (the hash of what is being built, taken BEFORE the compilers read it: a source edited while they run must not be
        # recorded as built -- it was, once, and the next build() then skipped a stale library)
        building = _source_hash()
        out = _build_locked(force, verbose)
        with open(STAMP, "w") as fh:
            fh.write(building + "\n")
        return out


def _build_locked(force, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = _flags()
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    compiles = []
    digests = {}
    for flavour, fl_flags in FLAVOURS.items():
        for unit, (files, unit_flags) in _units().items():
            deps = [os.path.join(CSRC, f) for f in files] + [HEADER]
            obj = _obj_path(unit, flavour)
            if obj in digests:
                continue   # the R flavour's row kernels are the double flavour's objects
            cmd = [hipcc] + fl_flags + flags + unit_flags + ["-c", os.path.join(CSRC, files[0]), "-o", obj]
            digests[obj] = _digest(cmd, deps)
            if force or not _fresh(obj, digests[obj]):
                if verbose:
                    print(" ".join(cmd))
                if os.path.exists(obj + ".stamp"):
                    os.remove(obj + ".stamp")
                compiles.append((cmd, obj, subprocess.Popen(cmd)))   # all stale objects compile side by side
    failed = None
    for cmd, obj, p in compiles:
        if p.wait() != 0:
            failed = failed or subprocess.CalledProcessError(p.returncode, cmd)
        else:
            _mark(obj, digests[obj])
    if failed:
        raise failed
    for flavour in FLAVOURS:
        out = lib_path(flavour)
        objs = [_obj_path(u, flavour) for u in _units()]
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
        digest = _digest(cmd + [digests[o] for o in objs], [])
        if force or not _fresh(out, digest):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            isa_guard(out)           # (before the library is marked as built: a violated invariant leaves it stale)
            _mark(out, digest)
    return lib_path(False), lib_path(True)


def isa_guard(lib):
    """scripts/isa_guard.py on a freshly linked library: the lane-team kernels must reach their cross-CU exchange through a call of the
    out-of-line team_sum_call, and that function must not spill while EXEC may be narrowed.  Raises on a violation;
    says so and goes on where llvm-objdump is missing.  POISMF_HIP_NO_ISA_GUARD=1 skips it (development builds of variants)."""
    if os.environ.get("POISMF_HIP_NO_ISA_GUARD") or "-DPMF_LANE_ONLY" in os.environ.get("POISMF_HIP_EXTRA_FLAGS", ""):
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmf_isa_guard", os.path.join(os.path.dirname(HERE), "scripts", "isa_guard.py"))
    guard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(guard)
    if not os.path.exists(guard.OBJDUMP):
        print(f"poismf_amd.build: no {guard.OBJDUMP}: ISA guard skipped for {os.path.basename(lib)}", file=sys.stderr)
        return
    lines = []
    if not guard.check_library(lib, lines.append):
        raise RuntimeError("ISA guard violated in " + lib + ":\n" + "\n".join(l for l in lines if ": info: " not in l))


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
