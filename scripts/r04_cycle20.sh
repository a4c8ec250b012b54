#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_moe.py -m gpu -q -x 2>&1 | tail -6
timeout 300 python scripts/moe_bench.py 1 2>/dev/null | tee gpurun_out/r04v_moe_mixtral_m1.json
timeout 300 python scripts/moe_bench.py 8 2>/dev/null | tee gpurun_out/r04v_moe_mixtral_m8.json
NS_MOE_GEMV_ROWS=0 timeout 300 python scripts/moe_bench.py 8 2>/dev/null | sed 's/^/[valu loop] /'
timeout 700 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
