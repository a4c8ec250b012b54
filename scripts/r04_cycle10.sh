#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04l_trace_warm.txt; : > $OUT
for sh in c4gu c4wq c4w2 c2gu; do
  NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_trace.so timeout 120 python scripts/gvs_trace.py $sh 2>&1 | tail -14 >> $OUT
done
cat $OUT
