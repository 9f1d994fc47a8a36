"""Builds the HIP shared libraries in-tree (poismf_amd/libpoismf_hip_{d,f}.so) with hipcc for gfx950.

    python -m poismf_amd.build            # both precisions
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "poismf_hip.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("poismf_hip.hip", "solvers.hpp", "row_eval.hpp", "wave_ops.hpp")]
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "poismf_hip.h"))


def lib_path(use_float):
    return os.path.join(HERE, "libpoismf_hip_f.so" if use_float else "libpoismf_hip_d.so")


def _stale(out):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in DEPS)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs = []
    for use_float in (False, True):
        out = lib_path(use_float)
        if not force and not _stale(out):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
               "-Wall", "-Wno-unused-function", "-Wno-pass-failed", SRC, "-o", out]
        if use_float:
            cmd.insert(1, "-DUSE_FLOAT")
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))  # the two precisions compile side by side
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    return lib_path(False), lib_path(True)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
