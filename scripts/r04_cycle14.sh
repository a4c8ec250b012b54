#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_load_path.py -m gpu -q -x 2>&1 | tail -8
timeout 600 python scripts/load_bench.py 4 3 2>/dev/null | tee gpurun_out/r04p_load_bench_70b_rank.json
