#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for nw in 0 4 8 16; do
  export NS_NW_PLAIN=$nw
  rm -rf gpurun_out/nwp
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/nwp -o a -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/nwp_bench.json 2>/dev/null
  echo "=== NS_NW_PLAIN=$nw $(python -c "import json; print(json.load(open('gpurun_out/nwp_bench.json'))['value'])")"
  python scripts/trace_summary.py gpurun_out/nwp/a_kernel_trace.csv smallm | tail -n +2
done
