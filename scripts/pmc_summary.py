#!/usr/bin/env python
"""Per-kernel-shape FETCH_SIZE from a rocprofv3 --pmc counter_collection.csv.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies the 128-byte requests
of wide coalesced streaming reads at 64 B -> double it.  rocprofv3 reports FETCH_SIZE in KiB."""
import csv, hashlib, json, os, sys, collections


def kernel_source_sha16():
    """hash of the decode kernel's sources: bench.py reads a committed counter summary back only while they are unchanged"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neural-speed_amd", "csrc")
    h = hashlib.sha256()
    for f in ("ns_gemv.hip", "ns_dev.h"):
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") != "FETCH_SIZE":
        continue
    name = r["Kernel_Name"]
    if "smallm" not in name and "gemv_kernel" not in name and "gemm" not in name and "gemvs" not in name:
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
# the summary bench.py reads back (roofline.traffic): the fused gate/up launch = gemv_kernel in dual mode (5th template
# argument 1), recorded with the kernel name and the launch grid so that a change of either invalidates it
gu = [(k, v) for k, v in out.items() if k.startswith("gemv_kernel<") and k.split("|")[0].rstrip(">").split(",")[4].strip() == "1"]
if gu and len(sys.argv) > 2:
    k, v = max(gu, key=lambda kv: kv[1]["calls"])
    _, g, w = k.split("|")
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 3 --warmup 1 (scripts/pmc_traffic.sh)",
               "correction": "FETCH_SIZE(KiB)*1024*2 (MI355X_MICROARCH.md HBM section: 128-B requests tallied at 64 B on gfx950)",
               "kernel_source_sha16": kernel_source_sha16(),
               "gate_up": dict(v, kernel="gemv_kernel", kernel_full=k.split("|")[0], grid=int(g) // int(w), workgroup=int(w)),
               "all": out}, open(sys.argv[2], "w"), indent=1)
    print("wrote", sys.argv[2])

# second mode (round 6): `pmc_summary.py <counter csv> --config4 <out json> <weight bytes per layer>` — the small-batch kernel's launches of
# BASELINE config 4 (bench.py --secondary-only): counter traffic of a layer's launches against the layer's algorithmic weight bytes.  A layer = one fused QKV
# launch (mode 2), one gate / up launch (mode 1), two plain launches (WO, down: mode 0) and the down projection's finalize; the output projection is a plain launch
# too (one per pass): the plain launches beyond two per layer are the largest ones and are left out.
if len(sys.argv) > 4 and sys.argv[2] == "--config4":
    layer_bytes = float(sys.argv[4])
    per = collections.defaultdict(list)   # kernel (with template arguments) -> bytes of every dispatch
    for r in rows:
        if r.get("Counter_Name") != "FETCH_SIZE" or "gemvs" not in r["Kernel_Name"]:
            continue
        per[r["Kernel_Name"].split("(")[0].replace("void ns::", "")].append(float(r["Counter_Value"]) * 1024 * 2)
    mode = lambda k: k.rstrip(">").split(",")[-1].strip() if "gemvs_kernel<" in k else "fin"
    n_layers = sum(len(v) for k, v in per.items() if mode(k) == "1")
    total, detail = 0.0, {}
    for k, v in per.items():
        v = sorted(v)
        if mode(k) == "0" and len(v) > 2 * n_layers:
            v = v[:2 * n_layers]   # (the output projection's launches: the largest plain ones)
        total += sum(v)
        detail[k] = {"dispatches_counted": len(v), "hbm_bytes_corrected_avg": sum(v) / max(1, len(v))}
    per_layer = total / max(1, n_layers)
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --secondary-only (scripts/pmc_traffic.sh)",
               "correction": "FETCH_SIZE(KiB)*1024*2 (MI355X_MICROARCH.md HBM section: 128-B requests tallied at 64 B on gfx950)",
               "layers_counted": n_layers, "config4_layer_hbm_bytes_corrected": per_layer, "config4_layer_algorithmic_weight_bytes": layer_bytes,
               "ratio": per_layer / layer_bytes if layer_bytes else None, "launches": detail}, open(sys.argv[3], "w"), indent=1)
    print("config 4: counter traffic per layer %.2f MB / algorithmic %.2f MB = %.3f (%d layers)" % (per_layer / 1e6, layer_bytes / 1e6, per_layer / layer_bytes if layer_bytes else 0, n_layers))
