#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04e_trace.txt; : > $OUT
for sh in c4gu c2gu c4w2 c4wq; do
  NS_GVS_DEBUG=1 NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_trace.so timeout 120 python scripts/gvs_trace.py $sh 2>&1 | grep -v "^gemvs: m 8.*$" | tail -16 >> $OUT
  NS_GVS_DEBUG=1 timeout 120 python scripts/gvs_probe.py $sh 2>&1 | grep "gemvs:\|PROBE" | sort | uniq -c | sort -rn | head -3 >> $OUT
done
NS_GVS_DEBUG=1 timeout 120 python scripts/gvs_probe.py c2w2 2>&1 | grep "gemvs:\|PROBE" | sort | uniq -c | sort -rn | head -3 >> $OUT
cat $OUT
