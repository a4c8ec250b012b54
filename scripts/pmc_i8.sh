#!/bin/bash
# SQ counters of the int8-MFMA kernel of the int8-reference mode (own --pmc pass): where the wave cycles go
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmci
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmci -o g -- python scripts/i8_prefill_bench.py > gpurun_out/pmci_bench.json 2>gpurun_out/pmci_err.log
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmci/g_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "i8mfma" not in r["Kernel_Name"] and "aquant" not in r["Kernel_Name"] and "i8prep" not in r["Kernel_Name"]:
        continue
    import re
    key = (re.search(r"(i8mfma2?_kernel<[^>]*>|aquant\w+|i8prep_kernel)", r["Kernel_Name"]).group(1), r["Grid_Size"])
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print(key, "dispatches", len(next(iter(c.values()))))
    wc = m.get("SQ_WAVE_CYCLES", 1)
    for k, v in sorted(m.items()):
        print("   %-28s %14.0f  %6.3f of WAVE_CYCLES" % (k, v, v / wc))
PY
