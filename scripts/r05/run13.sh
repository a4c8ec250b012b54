#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05n
timeout 600 python -m pytest tests/test_gpu_moe.py -x -q > gpurun_out/r05n/pytest.txt 2>&1; tail -3 gpurun_out/r05n/pytest.txt
python scripts/moe_bench.py 512 2>/dev/null | tail -1
NS_MOE_GROUPED_ROWS=0 python scripts/moe_bench.py 512 2>/dev/null | tail -1
python scripts/moe_bench.py 2048 2>/dev/null | tail -1
