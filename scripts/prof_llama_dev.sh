#!/bin/bash
# kernels the reference's UNCHANGED llama graph launches when it runs DEVICE-RESIDENT (its own -DNS_SYCL switch on
# libns_hip.so's bestla_device_* set): the 22-layer test model of tests/tools/llama_model_worker.py, then a
# Llama-2-7B-shaped synthetic model with tokens/s
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/prof_llama_dev /tmp/llw_dev
NS_WORKER_CONT_BATCH=0 timeout 200 python tests/tools/llama_model_worker.py product /tmp/llw_dev auto 4 > gpurun_out/prof_llama_dev_product.out 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_llama_dev -o m -- python tests/tools/llama_model_worker.py device /tmp/llw_dev f32 4 /tmp/llw_dev/llama_q_product_4.bin > gpurun_out/prof_llama_dev.out 2> gpurun_out/prof_llama_dev.err
echo "rc=$?"; grep "device-resident\|OK" gpurun_out/prof_llama_dev.out | cut -c1-300
find gpurun_out/prof_llama_dev -name "*kernel_stats.csv" -exec cp {} gpurun_out/llama_dev_kernel_stats.csv \;
find gpurun_out/prof_llama_dev -name "*kernel_trace.csv" -delete
head -12 gpurun_out/llama_dev_kernel_stats.csv | cut -c1-160
timeout 500 python scripts/dev_llama7b.py device 24 512 2>gpurun_out/dev_llama7b.err | tee gpurun_out/dev_llama7b.json | tail -2
rm -rf gpurun_out/prof7b; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof7b -o m -- python scripts/dev_llama7b.py device 12 512 > gpurun_out/prof7b.out 2> gpurun_out/prof7b.err
find gpurun_out/prof7b -name "*kernel_stats.csv" -exec cp {} gpurun_out/llama7b_dev_kernel_stats.csv \;
find gpurun_out/prof7b -name "*kernel_trace.csv" -delete
head -14 gpurun_out/llama7b_dev_kernel_stats.csv | cut -c1-160
