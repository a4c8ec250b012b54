#!/bin/bash
# attention tests + the whole-token leg with the merge inside the launch (default) and as a second launch (NS_ATTN_INLAUNCH=0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_kvcache.py tests/test_gpu_decoder_layer.py tests/test_gpu_whole_token_7b.py -m gpu -q -x 2>&1 | tail -3
for v in 1 0 1 0; do echo "attn_inlaunch=$v"; NS_ATTN_INLAUNCH=$v timeout 300 python bench.py --full-token-only 2>/dev/null | cut -c1-200; done
rm -rf gpurun_out/attnp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/attnp -o a -- python bench.py --full-token-only > /dev/null 2>&1
python scripts/kstats.py $(find gpurun_out/attnp -name '*kernel_stats.csv' | head -1) attn_
