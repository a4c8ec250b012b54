#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r04d}
T0=$(date +%s)
timeout 300 python -m pytest tests/test_gpu_gemvs.py -m gpu -q -x > gpurun_out/${TAG}_pytest_gemvs.log 2>&1
echo "PYTEST gemvs exit $? after $(( $(date +%s) - T0 )) s"; tail -5 gpurun_out/${TAG}_pytest_gemvs.log
OUT=gpurun_out/${TAG}_probe.txt; : > $OUT
for nw in 8 11 13 15; do NS_GVS_WAVES=$nw timeout 120 python scripts/gvs_probe.py c4gu 2>/dev/null | grep PROBE >> $OUT; done
for nw in 8 12 15; do NS_GVS_WAVES=$nw timeout 120 python scripts/gvs_probe.py c2gu 2>/dev/null | grep PROBE >> $OUT; done
for s in 2 4 8; do NS_GVS_SLICES=$s timeout 120 python scripts/gvs_probe.py c4w2 2>/dev/null | grep PROBE >> $OUT; done
timeout 120 python scripts/gvs_probe.py c4wq 2>/dev/null | grep PROBE >> $OUT
timeout 120 python scripts/gvs_probe.py c2wo 2>/dev/null | grep PROBE >> $OUT
timeout 120 python scripts/gvs_probe.py c2w2 2>/dev/null | grep PROBE >> $OUT
cat $OUT
timeout 300 python scripts/config_bench.py > gpurun_out/${TAG}_config_bench.json 2>gpurun_out/${TAG}_config_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_config_bench.json'))
for k in ('config4_mistral7b_nf4_g128_batch8','config5_llama70b_q4_0_rank_of_tp8'):
    print(k, json.dumps(d[k]['graph_chain']), {n:v['us'] for n,v in d[k]['per_shape'].items()})
PY
