#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r04u}
timeout 300 python -m pytest tests/test_gpu_gemvs.py -m gpu -q -x 2>&1 | tail -6
OUT=gpurun_out/${TAG}_probe.txt; : > $OUT
for s in 2 4; do for f in 1 0; do NS_GVS_FINALIZE=$f NS_GVS_SLICES=$s timeout 120 python scripts/gvs_probe.py c4w2 2>/dev/null | grep PROBE >> $OUT; done; done
for s in 1 2 4; do NS_GVS_SLICES=$s timeout 120 python scripts/gvs_probe.py c4gu 2>/dev/null | grep PROBE >> $OUT; done
for s in 1 2 4; do NS_GVS_SLICES=$s timeout 120 python scripts/gvs_probe.py c2gu 2>/dev/null | grep PROBE >> $OUT; done
for s in 1 2 4; do NS_GVS_SLICES=$s timeout 120 python scripts/gvs_probe.py c4wq 2>/dev/null | grep PROBE >> $OUT; done
for s in 2 4; do NS_GVS_SLICES=$s timeout 120 python scripts/gvs_probe.py c2w2 2>/dev/null | grep PROBE >> $OUT; done
cat $OUT
timeout 600 python bench.py --secondary-only 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for k in d: print(k, d[k]['us_per_layer'], d[k]['frac_of_8TBps'], d[k].get('tokens_per_s'), d[k]['parity_rel_l2_vs_oracle'])"
