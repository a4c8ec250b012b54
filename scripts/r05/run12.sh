#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05m
python scripts/r05/attn_layout_ab.py 2048 > gpurun_out/r05m/attn_layout_2048.json 2> gpurun_out/r05m/err.txt
cat gpurun_out/r05m/attn_layout_2048.json
python scripts/r05/attn_layout_ab.py 4096 > gpurun_out/r05m/attn_layout_4096.json 2>> gpurun_out/r05m/err.txt
cat gpurun_out/r05m/attn_layout_4096.json
timeout 300 python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | tail -2
