#!/bin/bash
# rocprofv3 kernel stats of the fused whole-token graph alone (ctx 2048): scripts/prof_full.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02o}
mkdir -p gpurun_out/$TAG; rm -rf gpurun_out/$TAG/prof
NS_FULL_ONLY_FUSED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -o $TAG -- python scripts/full_decode_bench.py ${CTX:-2048} > gpurun_out/$TAG/full_prof.json 2>/dev/null
find gpurun_out/$TAG/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/$TAG/full_token_kernel_stats.csv \;
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -size +20M -delete
grep -v "quantize_kernel\|reduce_kernel\|pack_codes\|scale_absmax\|repack_codes\|distribution_elementwise\|rocclr" gpurun_out/$TAG/full_token_kernel_stats.csv | head -16 | cut -c1-200
cat gpurun_out/$TAG/full_prof.json | tr -d '\n '; echo
