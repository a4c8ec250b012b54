#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python scripts/config_bench.py > gpurun_out/config_bench.json 2> gpurun_out/config_bench.err
echo rc=$?; tail -5 gpurun_out/config_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/config_bench.json"))
for k in ("config4_mistral7b_nf4_g128_batch8", "config5_llama70b_q4_0_rank_of_tp8"):
    print(k, {a: b for a, b in d[k].items() if a != "per_shape"})
PY
