"""rocprofv3 --kernel-trace (+ --memory-copy-trace) of scripts/dev_llama7b.py device ... -> (1) the prompt evaluation's timeline: span, busy time, gaps over 30 us,
time per kernel; (2) one replayed token, launch by launch.  Usage: route_timeline.py <kernel_trace.csv> [<memory_copy_trace.csv>]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
if len(sys.argv) > 2:
    try:
        cp = list(csv.DictReader(open(sys.argv[2])))
        for r in cp:
            r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            r['Kernel_Name'] = 'COPY %s %s B' % (r.get('Direction', ''), r.get('Bytes', r.get('Size', '')))
        rows += cp
    except Exception as ex:  # noqa: BLE001
        print('no copy trace:', ex)
rows.sort(key=lambda r: r['s'])
names = [r['Kernel_Name'] for r in rows]
i0 = next(i for i, n in enumerate(names) if 'gemm3_kernel' in n)
i1 = next((i for i, n in enumerate(names) if 'route_count_kernel' in n), len(rows))
while i0 > 0 and rows[i0]['s'] - rows[i0 - 1]['e'] < 3_000_000 and any(t in names[i0 - 1] for t in ('COPY', 'norm', 'copyBuffer', 'rope_cos_sin')):
    i0 -= 1
seg = rows[i0:i1]
t0 = seg[0]['s']
# the prompt ends with the copy of its logits: the first long quiet stretch behind it is the reference building the next token's graph
end = len(seg)
for i in range(1, len(seg)):
    if 'gemv_kernel' in seg[i]['Kernel_Name'] and seg[i]['s'] - seg[i - 1]['e'] > 200_000:
        end = i
        break
seg = seg[:end]
busy = sum(r['e'] - r['s'] for r in seg)
print("prompt evaluation (1500 tokens): %d launches / copies, span %.2f ms, busy %.2f ms" % (len(seg), (seg[-1]['e'] - t0) / 1e6, busy / 1e6))
prev = t0
agg = defaultdict(lambda: [0, 0])
for r in seg:
    gap = r['s'] - prev
    if gap > 30_000:
        print("  at %8.2f ms: gap %8.1f us before %s" % ((r['s'] - t0) / 1e6, gap / 1e3, r['Kernel_Name'][:100]))
    prev = max(prev, r['e'])
    k = r['Kernel_Name'].split('(')[0][:80]
    agg[k][0] += 1
    agg[k][1] += r['e'] - r['s']
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %5d x %9.1f us = %8.2f ms  %s" % (n, t / n / 1e3, t / 1e6, k))
idx = [i for i, n in enumerate(names) if 'route_count_kernel' in n]
if len(idx) >= 3:
    a, b = idx[-3], idx[-2]
    ta = rows[a]['s']
    print("\none replayed token at ~1560 cached positions (under the profiler every launch is serialised: durations, not overlap):")
    pe = ta
    for r in rows[a:b]:
        print("%9.2f us  gap %7.2f  dur %7.2f  %s" % ((r['s'] - ta) / 1e3, (r['s'] - pe) / 1e3, (r['e'] - r['s']) / 1e3, r['Kernel_Name'][:100]))
        pe = r['e']
