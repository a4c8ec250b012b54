#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for cfg in "1 0" "1 1" "0 0" "0 1"; do
  set -- $cfg
  if [ "$2" = "1" ]; then export NS_DEC_NO_ROUND_BARRIER=1; else unset NS_DEC_NO_ROUND_BARRIER; fi
  r=$(NS_DEC_CONTIG=$1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])")
  echo "contig $1 no_round_barrier $2 -> $r"
done
