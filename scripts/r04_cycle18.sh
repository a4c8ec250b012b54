#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_gemm3.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "prefill or qkv" 2>&1 | tail -6
echo "tests done after $(( $(date +%s) - T0 )) s"
for lib in default m16 default m16; do
  if [ $lib = default ]; then unset NS_LIB_PATH; else export NS_LIB_PATH=$GRAFT_REPO_ROOT/variants/libns_hip_$lib.so; fi
  timeout 300 python scripts/gemm_shapes_bench.py 2>/dev/null | tail -8 | sed "s/^/[$lib] /"
done
