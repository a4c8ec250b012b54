#!/bin/bash
cd /root/repo
for i in 1 2 3; do
NS_BENCH_PREFILL_ROPE_IN_QKV=0 python bench.py --full-token-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rope + append as a launch ', d['full_prefill'])"
python bench.py --full-token-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in the QKV GEMM epilogue  ', d['full_prefill'])"
done
