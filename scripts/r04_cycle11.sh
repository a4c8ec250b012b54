#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r04m}
OUT=gpurun_out/${TAG}_trace_warm.txt; : > $OUT
for sh in c4gu c4wq; do
  NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_trace.so timeout 120 python scripts/gvs_trace.py $sh 2>&1 | tail -14 >> $OUT
done
for sh in c4gu c2gu c4wq c4w2 c2w2; do timeout 120 python scripts/gvs_probe.py $sh 2>&1 | grep "PROBE" >> $OUT; done
cat $OUT
timeout 300 python -m pytest tests/test_gpu_gemvs.py -m gpu -q -x 2>&1 | tail -2
timeout 600 python bench.py --secondary-only 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for k in d: print(k, d[k]['us_per_layer'], d[k]['frac_of_8TBps'], d[k].get('tokens_per_s'), d[k]['parity_rel_l2_vs_oracle'])"
