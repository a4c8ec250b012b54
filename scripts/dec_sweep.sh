#!/bin/bash
# decode_kernel launch-shape sweep (diagnostics): workgroups x waves
cd $GRAFT_REPO_ROOT
for cfg in "256 12" "256 8" "512 6" "512 8" "768 4" "1024 4" "512 4" "256 6"; do
  set -- $cfg
  r=$(NS_DEC_GRID=$1 NS_DEC_NW=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])")
  echo "grid $1 nw $2 -> $r"
done
