#!/bin/bash
# usage: scripts/build_variant.sh <name> <flavours, e.g. f or "d f"> [extra hipcc flags...]
# A copy of the sources under _ab/<name>/ (git-ignored; travels with gpurun) built with extra flags, the named flavours only:
# development builds (-DPMF_PROBE, experiments behind macros) next to the product build, without rebuilding the latter.
# Run the copy's scripts with its own path: python3 _ab/<name>/scripts/probes/probe_lane.py
set -e
cd "$(dirname "$0")/.."
NAME=$1; FL=$2; shift 2
D=_ab/$NAME
mkdir -p $D
python3 - "$D" <<'PY'
import os, shutil, sys
d = sys.argv[1]
skip = shutil.ignore_patterns("*.o", "*.so", "*.stamp", ".build_*", "__pycache__")
for sub in ("poismf_amd", "include", "scripts"):
    # (objects already built in the copy stay: their stamps decide what is stale)
    shutil.copytree(sub, os.path.join(d, sub), ignore=skip, dirs_exist_ok=True)
PY
python3 - "$D" "$FL" <<'PY'
import re, sys
d, fl = sys.argv[1], sys.argv[2].split()
p = d + "/poismf_amd/build.py"
s = open(p).read()
all_fl = {"d": [], "f": ["-DUSE_FLOAT"], "r": ["-D_FOR_R"]}
s = re.sub(r"FLAVOURS = \{.*?\}", "FLAVOURS = " + repr({k: all_fl[k] for k in fl}), s, count=1)
open(p, "w").write(s)
PY
cd $D
POISMF_HIP_EXTRA_FLAGS="$*" python3 - <<'PY'
from poismf_amd import build
build._build_locked(False, False)
print("built", [build.lib_path(f) for f in build.FLAVOURS])
PY
