#!/bin/bash
# round 4, second GPU cycle: where does gemvs_kernel's time go — ablation builds + ring depth, true kernel durations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04b_ablation.txt; : > $OUT
for shape in c4gu c4i4gu c2gu; do
  for lib in default abl1 abl2 abl3 pf2 pf6 pf8; do
    if [ $lib = default ]; then unset NS_LIB_PATH; else export NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_$lib.so; fi
    for nw in 8 12; do
      NS_GVS_WAVES=$nw timeout 120 python scripts/gvs_probe.py $shape 2>/dev/null | grep PROBE >> $OUT
    done
  done
done
unset NS_LIB_PATH
for s in 2 4; do NS_GVS_SLICES=$s timeout 120 python scripts/gvs_probe.py c4w2 2>/dev/null | grep PROBE >> $OUT; done
for s in 2 4; do NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_abl3.so NS_GVS_SLICES=$s timeout 120 python scripts/gvs_probe.py c4w2 2>/dev/null | grep PROBE >> $OUT; done
NS_GVS=0 timeout 120 python scripts/gvs_probe.py c4gu 2>/dev/null | grep PROBE >> $OUT
NS_GVS=0 timeout 120 python scripts/gvs_probe.py c2gu 1 2>/dev/null | grep PROBE >> $OUT
NS_GVS=0 timeout 120 python scripts/gvs_probe.py c2gu 8 2>/dev/null | grep PROBE >> $OUT
cat $OUT
# true kernel durations (rocprofv3 kernel trace) for the default build
for shape in c4gu c4w2 c2gu; do
  rm -rf gpurun_out/kt
  NS_GVS_WAVES=8 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o t -- python scripts/gvs_probe.py $shape > /dev/null 2>&1
  echo "== kernel stats $shape"; python scripts/kstats.py $(find gpurun_out/kt -name '*kernel_stats.csv' | head -1) gemvs
done
