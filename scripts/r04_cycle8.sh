#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04i_pf_waves.txt; : > $OUT
for shape in c4gu c2gu; do
  for lib in default pf4 pf6 abl3 pf4abl3 pf6abl3; do
    if [ $lib = default ]; then unset NS_LIB_PATH; else export NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_$lib.so; fi
    for nw in 4 6 8 10 12 15; do
      NS_GVS_WAVES=$nw timeout 120 python scripts/gvs_probe.py $shape 2>/dev/null | grep PROBE | sed "s/knobs=.*: /nw=$nw: /" >> $OUT
    done
  done
done
cat $OUT
