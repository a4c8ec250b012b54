#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04h_ablation.txt; : > $OUT
for shape in c4gu c2gu c4wq; do
  for lib in default abl1 abl2 abl3; do
    if [ $lib = default ]; then unset NS_LIB_PATH; else export NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_$lib.so; fi
    timeout 120 python scripts/gvs_probe.py $shape 2>/dev/null | grep PROBE >> $OUT
  done
done
cat $OUT
