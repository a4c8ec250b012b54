#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (csv): for the steady-state tail of the run, GPU busy time, idle gaps between
consecutive kernels and launches per second — tells a launch-bound (host) loop from a kernel-bound one.
usage: trace_gaps.py <kernel_trace.csv> [tail_fraction]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = rows[int(len(rows) * (1 - frac)):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
gaps_sorted = sorted(gaps)
print("kernels %d  span %.1f us  busy %.1f us (%.1f%%)  per kernel: span %.2f us, busy %.2f us, gap median %.2f us mean %.2f us p90 %.2f us"
      % (len(rows), span / 1e3, busy / 1e3, 100.0 * busy / span, span / 1e3 / len(rows), busy / 1e3 / len(rows),
         gaps_sorted[len(gaps) // 2] / 1e3, sum(gaps) / 1e3 / len(gaps), gaps_sorted[int(len(gaps) * 0.9)] / 1e3))
by = collections.defaultdict(lambda: [0, 0, 0])
for i, (s, e, n) in enumerate(rows[:-1]):
    by[n][0] += 1; by[n][1] += e - s; by[n][2] += gaps[i]
for n, (c, d, g) in sorted(by.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:14]:
    print("%-60s calls %5d  avg dur %6.2f us  avg gap after %6.2f us" % (n, c, d / c / 1e3, g / c / 1e3))
