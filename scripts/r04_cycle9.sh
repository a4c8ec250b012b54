#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r04k}
timeout 300 python -m pytest tests/test_gpu_gemvs.py -m gpu -q -x > gpurun_out/${TAG}_pytest_gemvs.log 2>&1
echo "PYTEST gemvs exit $?"; tail -3 gpurun_out/${TAG}_pytest_gemvs.log
T0=$(date +%s)
NS_GVS_DEBUG=1 timeout 600 python bench.py --secondary-only > gpurun_out/${TAG}_secondary.json 2>gpurun_out/${TAG}_secondary.err
echo "SECONDARY exit $? after $(( $(date +%s) - T0 )) s"; cat gpurun_out/${TAG}_secondary.json; sort gpurun_out/${TAG}_secondary.err | uniq -c | sort -rn | head -20
