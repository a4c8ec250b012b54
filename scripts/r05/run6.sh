#!/bin/bash
set -x
mkdir -p gpurun_out/r05f /tmp/w5
cd /root/repo
python tests/tools/llama_model_worker.py product /tmp/w5 auto 4 - llama > gpurun_out/r05f/product.txt 2>&1
NS_ROUTE_DUMP=48 NS_WORKER_N_NEW=6 python tests/tools/llama_model_worker.py device /tmp/w5 f32 4 /tmp/w5/llama_q_product_4.bin llama > gpurun_out/r05f/device.txt 2>&1
grep -n "route plan" gpurun_out/r05f/device.txt | cut -c1-260
