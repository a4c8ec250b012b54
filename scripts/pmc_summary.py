#!/usr/bin/env python
"""Per-kernel-shape FETCH_SIZE from a rocprofv3 --pmc counter_collection.csv.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies the 128-byte requests
of wide coalesced streaming reads at 64 B -> double it.  rocprofv3 reports FETCH_SIZE in KiB."""
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") != "FETCH_SIZE":
        continue
    name = r["Kernel_Name"]
    if "smallm" not in name and "decode_kernel" not in name and "gemm" not in name:
        continue
    short = name.split("(")[0].replace("void ns::", "")
    key = (short, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
    agg[key].append(float(r["Counter_Value"]))
out = {}
print("%-56s %8s %6s %6s %14s %14s" % ("kernel", "grid", "wg", "calls", "FETCH_KiB", "HBM_MB(x2)"))
for (k, g, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    avg = sum(v) / len(v)
    mb = avg * 1024 * 2 / 1e6
    print("%-56s %8s %6s %6d %14.1f %14.2f" % (k[:56], g, w, len(v), avg, mb))
    out["%s|%s|%s" % (k, g, w)] = {"calls": len(v), "fetch_size_kib_avg": avg, "hbm_bytes_corrected": avg * 1024 * 2}
json.dump(out, open(sys.argv[1].replace(".csv", "_summary.json"), "w"), indent=1)
