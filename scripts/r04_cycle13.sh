#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_load_path.py tests/test_gpu_gemvs.py -m gpu -q -x 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_llama_model.py -m gpu -q -x 2>&1 | tail -5
T0=$(date +%s)
NS_LOAD_STATS=1 timeout 600 python scripts/dev_llama7b.py device 16 512 > gpurun_out/r04o_dev7b.txt 2>&1
echo "dev7b exit $? after $(( $(date +%s) - T0 )) s"; grep -v "^llama\|^model\|^ne_\|^init" gpurun_out/r04o_dev7b.txt | tail -12
NS_LOAD_OWN_ALLOC=1 NS_LOAD_STATS=1 timeout 600 python scripts/dev_llama7b.py device 16 512 2>&1 | grep "load\|route" | tail -4
