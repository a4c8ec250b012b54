#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for g in 1 2; do
  for nw in 0 8 15; do
    NS_GVS=$g NS_GVS_WAVES=$nw timeout 200 python bench.py --chain-only --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('gvs=$g waves=$nw tok/s', d['value'], 'ms', d['ms_per_step'])"
  done
done
