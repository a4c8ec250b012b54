#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_attention.py tests/test_gpu_kvcache.py tests/test_gpu_decoder_layer.py tests/test_gpu_int8_mode.py -x -q 2>&1 | tail -4
timeout 120 python scripts/attn_prefill_bench.py 2>&1 | tail -1 | tee gpurun_out/attn_prefill_vgprform.json
