#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid, block) with avg/median/min durations."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    short = name.replace("ns::(anonymous namespace)::", "").replace("void ns::", "").split("(")[0][-48:]
    key = (short, r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"])
    d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("%-50s %9s %5s %5s %6s %6s %9s %9s %9s" % ("kernel", "grid", "wg", "vgpr", "lds", "calls", "avg_us", "med_us", "min_us"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-50s %9s %5s %5s %6s %6d %9.2f %9.2f %9.2f" % (k[0], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3))
