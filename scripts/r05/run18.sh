#!/bin/bash
cd /root/repo
for i in 1 2; do
python bench.py --full-token-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge launch   ', d['full_token'])"
NS_ATTN_INLAUNCH=1 python bench.py --full-token-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge in launch', d['full_token'])"
done
