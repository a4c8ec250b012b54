#!/bin/bash
# first GPU visit of round 5: the GPU suite's quick subset, the bench line, the prefill probe (default / stores skipped)
set -x
mkdir -p gpurun_out/r05a
cd /root/repo
python scripts/r05/prefill_probe.py 2048 40 > gpurun_out/r05a/probe_default.json 2> gpurun_out/r05a/probe_default.err
NS_G3_DIAG=1 python scripts/r05/prefill_probe.py 2048 40 --no-lib > gpurun_out/r05a/probe_nostore.json 2>> gpurun_out/r05a/probe_default.err
python -m pytest tests/test_gpu_device_lazy.py tests/test_gpu_gemm3.py tests/test_gpu_load_path.py tests/test_gpu_gemvs.py -x -q > gpurun_out/r05a/pytest_subset.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err
tail -3 gpurun_out/r05a/pytest_subset.txt
cat gpurun_out/r05a/probe_default.json gpurun_out/r05a/probe_nostore.json
