#!/bin/bash
# round 5, second GPU visit: the new gemm3 epilogue / gate-up pairs — tests, per-shape probe (wide on / off), the bench line
set -x
mkdir -p gpurun_out/r05b
cd /root/repo
python -m pytest tests/test_gpu_gemm3.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r05b/pytest_gemm.txt 2>&1
tail -5 gpurun_out/r05b/pytest_gemm.txt
python scripts/r05/prefill_probe.py 2048 40 --no-lib > gpurun_out/r05b/probe_wide.json 2> gpurun_out/r05b/probe.err
NS_G3_WIDE=0 python scripts/r05/prefill_probe.py 2048 40 --no-lib > gpurun_out/r05b/probe_perwave.json 2>> gpurun_out/r05b/probe.err
cat gpurun_out/r05b/probe_wide.json gpurun_out/r05b/probe_perwave.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05b/bench.json').read().strip().splitlines()[-1])
c=d['config']
print(d['value'], c.get('tokens_per_s_median_of_5_more_batches'), c.get('full_token_tokens_per_s'), c.get('prefill_m2048_tflops'), c.get('prefill_m2048_tflops_int8w'), c.get('prefill_m2048_tflops_ref_int8_semantics'))
print(c['prefill_m2048_detail']); print(c['full_prefill'])
P
tail -5 gpurun_out/r05b/bench.err
