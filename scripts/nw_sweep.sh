#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for nw in 0 2 4 8 16; do
  export NS_NW=$nw
  rm -rf gpurun_out/abl
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl -o a -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/abl_bench.json 2>/dev/null
  echo "=== NS_NW=$nw  $(python -c "import json; d=json.load(open('gpurun_out/abl_bench.json')); print(d['value'], 'tok/s')")"
  python scripts/trace_summary.py gpurun_out/abl/a_kernel_trace.csv smallm | tail -n +2
done
