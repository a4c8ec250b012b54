#!/bin/bash
# whole-token cycle: new parity tests, whole-token bench at ctx 128/512/2048, rocprofv3 kernel stats of a ctx-2048 run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02n}
mkdir -p gpurun_out/$TAG
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_norm_link.py tests/test_gpu_attention.py "tests/test_gpu_configs.py::test_config2_llama7b_q4_0_decode_full_size" -x -q 2>&1 | tail -15
echo "TESTS done after $(( $(date +%s) - T0 )) s"
timeout 300 python scripts/full_decode_bench.py 128 512 2048 > gpurun_out/$TAG/full_token.json 2>gpurun_out/$TAG/full_err.log
echo "FULL exit $?"; cat gpurun_out/$TAG/full_token.json; tail -5 gpurun_out/$TAG/full_err.log
rm -rf gpurun_out/$TAG/prof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -o $TAG -- python scripts/full_decode_bench.py 2048 > /dev/null 2>&1
find gpurun_out/$TAG/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/$TAG/full_token_kernel_stats.csv \;
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -size +20M -delete
head -14 gpurun_out/$TAG/full_token_kernel_stats.csv | cut -c1-220
timeout 300 python bench.py --chain-only 2>&1 | tail -1 | cut -c1-600
echo "DONE after $(( $(date +%s) - T0 )) s"
