"""Builds the HIP shared libraries in-tree (poismf_amd/libpoismf_hip_{d,f}.so) with hipcc for gfx950.

    python -m poismf_amd.build [--force] [-v]

Two translation units per precision (the row kernels + host side, and the rocPRIM-based COO conversion) are
compiled to object files side by side and linked; only stale objects are rebuilt.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(os.path.dirname(HERE), "include", "poismf_hip.h")
UNITS = {
    "poismf_hip": ["poismf_hip.hip", "solvers.hpp", "row_eval.hpp", "reg_eval.hpp", "wave_ops.hpp"],
    "coo_convert": ["coo_convert.hip"],
    "serve": ["serve.hip"],
}


def lib_path(use_float):
    return os.path.join(HERE, "libpoismf_hip_f.so" if use_float else "libpoismf_hip_d.so")


def _obj_path(unit, use_float):
    return os.path.join(CSRC, f"{unit}_{'f' if use_float else 'd'}.o")


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
             "-Wno-pass-failed"]
    flags += os.environ.get("POISMF_HIP_EXTRA_FLAGS", "").split()   # development: e.g. -DPMF_REG_G=8, -DPMF_TIMING
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    compiles = []
    for use_float in (False, True):
        for unit, files in UNITS.items():
            deps = [os.path.join(CSRC, f) for f in files] + [HEADER]
            obj = _obj_path(unit, use_float)
            if force or _stale(obj, deps):
                cmd = [hipcc] + (["-DUSE_FLOAT"] if use_float else []) + flags + ["-c", os.path.join(CSRC, files[0]), "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                compiles.append((cmd, subprocess.Popen(cmd)))   # all stale objects compile side by side
    for cmd, p in compiles:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    for use_float in (False, True):
        out = lib_path(use_float)
        objs = [_obj_path(u, use_float) for u in UNITS]
        if force or _stale(out, objs):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return lib_path(False), lib_path(True)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
