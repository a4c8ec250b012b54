#!/usr/bin/env python
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
for b in blocks:
    name = b.split("\n")[0].split(" [-R")[0].strip()
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(.*", "", dem).replace("void ns::", "")
    if flt and flt not in dem:
        continue
    g = lambda k: (re.search(r"\s" + re.escape(k) + r": (\d+)", b) or [None, "?"])[1]
    print("%-60s vgpr %4s agpr %4s sgpr %4s scratch %5s occ %2s lds %6s" % (
        dem, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
        g("LDS Size [bytes/block]")))
