#!/bin/bash
# kernel-trace stats of the auxiliary benches (attention decode, prefill GEMM, secondary configs)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/aux
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/aux -o attn -- python scripts/attn_bench.py > gpurun_out/aux_attn.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/aux -o gemm -- python scripts/prefill_bench.py > gpurun_out/aux_gemm.json 2>/dev/null
python - <<'PY'
import csv
for tag in ("attn", "gemm"):
    rows = list(csv.DictReader(open("gpurun_out/aux/%s_kernel_stats.csv" % tag)))
    print("==", tag)
    for r in rows[:8]:
        print("  %-70s calls %6s avg_us %10.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
