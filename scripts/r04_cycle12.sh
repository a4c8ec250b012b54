#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04n_rows.txt; : > $OUT
for m in 2 3 4 6 12 16; do
  for sh in c2wo c2gu c2w2 c4wq c4gu c4w2; do
    for g in 0 1; do
      NS_GVS=$g timeout 120 python scripts/gvs_probe.py $sh $m 2>&1 | grep "PROBE" | sed "s/lib=default knobs=.*: /gvs=$g /" >> $OUT
    done
  done
done
cat $OUT
