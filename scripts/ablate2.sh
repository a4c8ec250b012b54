#!/bin/bash
# compile-time ablation builds of the decode kernel (diagnostics): NS_ABLATE 1 = no dequant/MFMA, 2 = no scales, 4 = no A staging
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ab in 0 1 3 7; do
  rm -f neural-speed_amd/csrc/ns_kernels.o
  make -C neural-speed_amd/csrc -j8 EXTRA=-DNS_ABLATE=$ab > /dev/null 2>&1
  rm -rf gpurun_out/abl
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl -o a -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/abl_bench.json 2>/dev/null
  echo "=== NS_ABLATE=$ab  $(python -c "import json; d=json.load(open('gpurun_out/abl_bench.json')); print(d['value'], 'tok/s')")"
  python scripts/trace_summary.py gpurun_out/abl/a_kernel_trace.csv smallm | tail -n +2
done
