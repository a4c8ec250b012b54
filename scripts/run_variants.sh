#!/bin/bash
# run the decode bench under rocprofv3 for each variants/libns_hip_<name>.so given on the command line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for name in "$@"; do
  export NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_$name.so
  rm -rf gpurun_out/var
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/var -o a -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/var_bench.json 2>/dev/null
  echo "=== $name  $(python -c "import json; d=json.load(open('gpurun_out/var_bench.json')); print(d['value'], 'tok/s')")"
  python scripts/trace_summary.py gpurun_out/var/a_kernel_trace.csv "smallm\|decode" | tail -n +2
done
