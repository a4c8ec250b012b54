#!/bin/bash
# kernels the reference's UNCHANGED llama model code launches on libns_hip.so (quantize through its driver, load, generate)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/prof_llama
NS_WORKER_CONT_BATCH=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_llama -o m -- python tests/tools/llama_model_worker.py product /tmp/llw_prof auto 4 > gpurun_out/prof_llama.out 2> gpurun_out/prof_llama.err
echo "rc=$?"; grep "^llama\|OK" gpurun_out/prof_llama.out | cut -c1-300
find gpurun_out/prof_llama -name "*kernel_stats.csv" -exec cp {} gpurun_out/llama_model_kernel_stats.csv \;
find gpurun_out/prof_llama -name "*kernel_trace.csv" -delete
head -14 gpurun_out/llama_model_kernel_stats.csv | cut -c1-160
