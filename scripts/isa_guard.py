#!/usr/bin/env python3
"""isa_guard.py [libpoismf_hip_X.so ...] -- the ISA-level invariants of the multi-CU team kernels, checked on the BUILT libraries.

DESIGN.md 4.8 records a hazard seen on hardware in round 5: inside the 512-register team kernels a value that is live in ALL lanes was
spilled at a point where only some lanes were active (a scratch_store under a narrowed EXEC mask) and reloaded after the lanes had
reconverged -- the lanes that were switched off at the store came back with whatever that scratch slot held before.  The source-level rule
that keeps the exchange code clear of it is "no lane-dependent branch in team_sum, and team_sum called out of line".  This script turns the
rule into something a build can fail on:

  1. `team_sum_call` exists as a function of its own in every code object that has a lane-team kernel (the 512-register instances), and those
     kernels reach it through a call (s_swappc_b64) -- i.e. the compiler has not inlined the exchange into the 512-register body;
  2. in `team_sum_call` itself no scratch_store executes while EXEC may be narrowed: between an `s_and_saveexec_b64` (or any other write of
     EXEC that is not `s_mov_b64 exec, -1`) and the instruction that puts the saved mask back.
The giant-row kernels inline the exchange by design (they are not register-bound); their scratch-store counts are printed for the record.
scripts/probes/spill_divergent.hip is the attempt to reproduce the hazard in isolation (result in DESIGN.md 4.8).

The scan is linear over the disassembly (llvm-objdump -d of the gfx950 code objects found in the library's offload bundles), with a stack
of saved-mask registers: a heuristic that errs on the side of reporting.  Exit code 0 = invariants hold, 1 = violated (the offending
instructions are printed), 2 = could not run (no llvm-objdump).  `poismf_amd.build.build()` runs it after linking; tests/test_isa_guard.py
runs it on the in-tree libraries and on a synthetic listing that must fail.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
LANE_TEAM_KERNEL = re.compile(r"half_sweep_lane_team_kernel")
GIANT_KERNEL = re.compile(r"half_sweep_giant_kernel")
EXCHANGE = re.compile(r"team_sum_call")


def code_objects(path):
    """the gfx950 ELF images inside the clang offload bundles of a host object / shared library"""
    blob = open(path, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        o = m.start()
        (n,) = struct.unpack_from("<Q", blob, o + 24)
        p = o + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + ts].decode(errors="replace")
            p += ts
            if "gfx950" in triple and size > 0:
                out.append(blob[o + off:o + off + size])
    return out


def disassemble(image):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        return subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "-C", f.name], capture_output=True, text=True, check=True).stdout


def functions(listing):
    """{demangled name: [instruction text, ...]} of one disassembly"""
    funcs, cur = {}, None
    for line in listing.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is not None and line.startswith((" ", "\t")) and line.strip():
            ins = line.split("//")[0].strip()
            if ins:
                cur.append(ins)
    return funcs


_EXEC_DST = re.compile(r"^(s_\w+)\s+exec(_lo|_hi)?\b")


def narrowed_scratch_stores(insns):
    """scratch_store instructions met while EXEC may be narrower than at function entry: [(index, text, why)]"""
    saved = []   # registers holding a saved EXEC (innermost last); "?" = EXEC written in some other way
    bad = []
    for i, ins in enumerate(insns):
        op = ins.split()[0]
        if op.startswith("s_and_saveexec") or op.startswith("s_or_saveexec") or op.startswith("s_andn2_saveexec") or op.startswith("s_xor_saveexec"):
            saved.append(ins.split()[1].rstrip(","))
            continue
        m = _EXEC_DST.match(ins)
        if m:
            args = [a.strip() for a in ins.split(None, 1)[1].split(",")]
            src = args[1:]
            if op.startswith("s_mov") and src == ["-1"]:
                saved.clear()
            elif (op.startswith("s_or_b") or op.startswith("s_mov_b")) and saved and any(s == saved[-1] for s in src):
                saved.pop()          # the innermost saved mask goes back
            elif op.startswith("s_or_b") and any(s in saved for s in src):
                while saved and saved[-1] not in src:
                    saved.pop()
                saved.pop()
            elif op.startswith("s_or_b"):
                pass                 # lanes only ever come back through an OR: never narrower than before
            else:
                saved.append("?")    # s_and / s_andn2 / s_xor on EXEC: a loop's or an else's mask
            continue
        if op.startswith("scratch_store") and saved:
            bad.append((i, ins, "EXEC saved in " + ",".join(saved)))
    return bad


def check_listing(listing, where, report):
    """Hard rules: (1) every lane-team kernel (512 registers + scratch) reaches the exchange through a CALL of an out-of-line team_sum_call;
    (2) team_sum_call itself never spills while EXEC may be narrowed.  The giant-row kernels inline the exchange (row_eval.hpp: they have
    registers to spare for it) -- for them the linear scan over tens of thousands of instructions cannot tell a loop's EXEC bookkeeping from a
    divergent region, so their spill counts are reported, not judged."""
    funcs = functions(listing)
    lane_team = {n: f for n, f in funcs.items() if LANE_TEAM_KERNEL.search(n)}
    giant = {n: f for n, f in funcs.items() if GIANT_KERNEL.search(n)}
    ok = True
    exch = {n: f for n, f in funcs.items() if EXCHANGE.search(n) and not LANE_TEAM_KERNEL.search(n) and not GIANT_KERNEL.search(n)}
    if lane_team and not exch:
        report(f"{where}: {len(lane_team)} lane-team kernel(s) but no out-of-line team_sum_call: the exchange was inlined")
        ok = False
    for name, insns in lane_team.items():
        if not any(i.startswith("s_swappc_b64") for i in insns):
            report(f"{where}: {name[:120]}: no call (s_swappc_b64) -- the exchange was inlined into the kernel")
            ok = False
    for name, insns in exch.items():
        for idx, ins, why in narrowed_scratch_stores(insns):
            report(f"{where}: {name[:120]}: instruction {idx}: {ins}   <- a spill while EXEC may be narrowed ({why})")
            ok = False
    for name, insns in list(lane_team.items()) + list(giant.items()):
        n_st = sum(i.startswith("scratch_store") for i in insns)
        report(f"{where}: info: {name[:110]}: {len(insns)} instructions, {n_st} scratch stores, {sum(i.startswith('s_swappc') for i in insns)} calls")
    return ok


def check_library(path, report=print):
    ok, n = True, 0
    for image in code_objects(path):
        listing = disassemble(image)
        n += 1
        ok = check_listing(listing, f"{os.path.basename(path)}[{n}]", report) and ok
    return ok


def main(argv):
    if not os.path.exists(OBJDUMP):
        print(f"isa_guard: no {OBJDUMP}")
        return 2
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = argv or [os.path.join(here, "poismf_amd", f"libpoismf_hip_{f}.so") for f in ("d", "f")]
    ok = True
    for lib in libs:
        ok = check_library(lib) and ok
    print("isa_guard:", "ok" if ok else "VIOLATED", "--", ", ".join(os.path.basename(p) for p in libs))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
