#!/bin/bash
# kernel-level ablation of the decode chain (diagnostics): per-shape kernel durations under NS_ABLATE / NS_NW settings
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 0" "3 0" "7 0" "2 0" "4 0" "0 4" "0 16"; do
  set -- $cfg
  export NS_ABLATE=$1 NS_NW=$2
  rm -rf gpurun_out/abl
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl -o a -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "=== NS_ABLATE=$1 NS_NW=$2"
  python scripts/trace_summary.py gpurun_out/abl/a_kernel_trace.csv smallm | tail -n +2
done
