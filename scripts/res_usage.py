#!/usr/bin/env python3
"""res_usage.py <file with -Rpass-analysis=kernel-resource-usage remarks> : one line per kernel: VGPRs AGPRs scratch occupancy LDS name"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
cur = {}
for line in txt.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        continue
    for key, tag in (("VGPRs", "v"), ("AGPRs", "a"), ("ScratchSize [bytes/lane]", "s"), ("Occupancy [waves/SIMD]", "o"), ("LDS Size [bytes/block]", "l")):
        m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
        if m and cur:
            cur[tag] = int(m.group(1))
            if tag == "l":
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                print(f"{cur.get('v', 0):4d} {cur.get('a', 0):4d} {cur.get('s', 0):5d} {cur.get('o', 0)} {cur['l']:7d} | {name[:150]}")
                cur = {}
