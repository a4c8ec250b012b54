#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r04f}
timeout 300 python -m pytest tests/test_gpu_gemvs.py -m gpu -q -x > gpurun_out/${TAG}_pytest_gemvs.log 2>&1
echo "PYTEST gemvs exit $?"; tail -3 gpurun_out/${TAG}_pytest_gemvs.log
OUT=gpurun_out/${TAG}_trace.txt; : > $OUT
for sh in c4gu c4w2 c4wq; do
  NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_trace.so timeout 120 python scripts/gvs_trace.py $sh 2>&1 | tail -14 >> $OUT
done
for sh in c4gu c2gu c4wq c2wo c2w2; do NS_GVS_DEBUG=1 timeout 120 python scripts/gvs_probe.py $sh 2>&1 | grep "gemvs:\|PROBE" | sort | uniq | head -3 >> $OUT; done
for s in 2 4 8; do NS_GVS_SLICES=$s timeout 120 python scripts/gvs_probe.py c4w2 2>/dev/null | grep PROBE >> $OUT; done
cat $OUT
timeout 300 python scripts/config_bench.py > gpurun_out/${TAG}_config_bench.json 2>gpurun_out/${TAG}_config_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_config_bench.json'))
for k in ('config4_mistral7b_nf4_g128_batch8','config5_llama70b_q4_0_rank_of_tp8'):
    print(k, json.dumps(d[k]['graph_chain']), {n:v['us'] for n,v in d[k]['per_shape'].items()})
PY
