#!/bin/bash
# prefill attention: parity tests, then the 64-row kernel of rounds 1-3 (NS_ATTN_MFMA2_ROWS huge) against the 128-row kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_kvcache.py -m gpu -q -x 2>&1 | tail -5
if [ "$1" = "all" ]; then
echo "64-row kernel"; NS_ATTN_MFMA2_ROWS=1000000 timeout 300 python scripts/attn_prefill_bench.py 512 1024 2048 4096 8192 2>&1 | tail -1
echo "128-row kernel, plain workgroup order"; NS_ATTN_NO_XCD_MAP=1 timeout 300 python scripts/attn_prefill_bench.py 512 1024 2048 4096 8192 2>&1 | tail -1
fi
echo "128-row kernel"; timeout 300 python scripts/attn_prefill_bench.py 512 1024 2048 4096 8192 2>&1 | tail -1
