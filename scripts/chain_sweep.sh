#!/bin/bash
# persistent chain kernel: polling cadence sweep (first sleep, gap; units of 512 cycles)
for pf in 2 6 12; do for pg in 1 2 6; do
  echo "poll_first=$pf gap=$pg"; NS_CHAIN_POLL_FIRST=$pf NS_CHAIN_POLL_GAP=$pg CHAIN_TRACE=1 timeout 120 scripts/ubench/chain_bench --layers 8 --modes 1 --reps 5 --chain 2>&1 | grep "persistent\|^op  [0-3]"
done; done
