#!/bin/bash
# decode-kernel parameter sweep on the GPU box: ring depth (library variants) x waves per workgroup (NS_GV_NW)
out=gpurun_out/${1:-sweep}; mkdir -p $out
for v in base pf4; do
  for nw in 0 4 8 16; do
    lib=""; [ $v != base ] && lib=variants/$v
    LD_LIBRARY_PATH=$lib NS_GV_NW=$nw timeout 120 scripts/ubench/chain_bench --modes 1 --reps 20 2>/dev/null | grep gemv2 | sed "s/^/$v nw=$nw /" | cut -c1-200 | tee -a $out/sweep.txt
  done
done
