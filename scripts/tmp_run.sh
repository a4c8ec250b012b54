#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
time (NS_WORKER_WATCHDOG_S=60 timeout 90 python tests/tools/llama_model_worker.py oracle /tmp/llw f16 2 tmp_llama_q.bin > gpurun_out/llw2.out 2> gpurun_out/llw2.err)
echo "rc=$?"
grep "^llama\|OK\|Timeout" gpurun_out/llw2.out gpurun_out/llw2.err | tail
