#!/usr/bin/env python3
"""print name / calls / average us of the kernels in a rocprofv3 --stats kernel_stats CSV (optionally filtered)"""
import csv, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get("Name") or r.get("KernelName") or ""
    if pat and pat not in name:
        continue
    calls = int(r.get("Calls", 0)); avg = float(r.get("AverageNs", 0)) / 1e3
    print("  %-70s calls %5d avg %8.2f us  min %.2f max %.2f" % (name[:70], calls, avg, float(r.get("MinNs", 0)) / 1e3, float(r.get("MaxNs", 0)) / 1e3))
