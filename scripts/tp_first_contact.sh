#!/bin/bash
# First contact with a multi-GPU MI355X node: everything the tensor-parallel layer has only ever run with ranks
# sharing ONE GPU, in one command, each step skipping cleanly when the node has fewer GPUs than it needs.
#   (a) litmus of ns_p2p.hip's drained-sc1 hand-off across devices: N peer-memory all-reduces with (sequence, element, rank)-tagged
#       payloads, exact sums, uneven load on the odd ranks               -> P2P_LITMUS_OK
#   (b) libns_hip.so's ns_tp_* layer + glue/parallel_context_hip.cpp over the REAL RCCL (no stand-in), one GPU per rank, at 2 / 4 / 8
#       ranks, against the ranks' sums                                    -> TP_NATIVE_OK
#   (c) bench.py --gpus {1,2,4,8}: tokens/s, all_reduce_us, comm_fraction per N (the driver computes scaling efficiency itself)
# Usage: scripts/tp_first_contact.sh [litmus calls, default 200000]      env: NS_FC_SKIP_BENCH=1 skips (c), NS_FC_OUT=dir for the logs
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
CALLS=${1:-200000}
OUT=${NS_FC_OUT:-gpurun_out/tp_first_contact}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NGPU=$(python - <<'P'
import torch
print(torch.cuda.device_count() if torch.cuda.is_available() else 0)
P
)
echo "tp_first_contact: $NGPU GPU(s) visible"
rc=0
run_ranks() {  # run_ranks <n> <port> <script> [args...]
  local n=$1 port=$2; shift 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" "$@"
}
# ---- (a) ----------------------------------------------------------------------------------------------------------------------
for n in 2 4 8; do
  if [ "$NGPU" -lt "$n" ]; then echo "(a) litmus, $n ranks: SKIP (needs $n GPUs)"; continue; fi
  NS_P2P_TIMEOUT_MS=20000 run_ranks "$n" $((29600 + n)) scripts/tp/p2p_litmus_worker.py "$CALLS" 4096 > "$OUT/litmus_$n.log" 2>&1
  if grep -q P2P_LITMUS_OK "$OUT/litmus_$n.log"; then grep P2P_LITMUS_OK "$OUT/litmus_$n.log"; else echo "(a) litmus, $n ranks: FAIL (see $OUT/litmus_$n.log)"; tail -5 "$OUT/litmus_$n.log"; rc=1; fi
done
# ---- (b) ----------------------------------------------------------------------------------------------------------------------
GLUE="$OUT/libpc_glue.so"
if [ "$NGPU" -ge 2 ]; then
  REFINC=${NS_REFERENCE_CORE:-/root/reference/neural_speed/core}
  if [ -f oracle/_ref/libpc_glue.so ]; then cp oracle/_ref/libpc_glue.so "$GLUE";
  elif [ -d "$REFINC" ]; then g++ -O2 -std=c++17 -fPIC -shared -I"$REFINC" -Iinclude glue/parallel_context_hip.cpp -o "$GLUE";
  else echo "(b): no prebuilt oracle/_ref/libpc_glue.so and no reference headers (NS_REFERENCE_CORE): SKIP"; GLUE=""; fi
fi
for n in 2 4 8; do
  if [ "$NGPU" -lt "$n" ]; then echo "(b) ns_tp over RCCL, $n ranks: SKIP (needs $n GPUs)"; continue; fi
  [ -z "$GLUE" ] && continue
  env -u NS_TP_RCCL_LIB -u NS_TP_ID_FILE NS_TP_WORKER_DEVICE=local NS_TP_WORKER_EXACT=0 NS_TP_RUN_ID="fc$$-$n" \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29620 + n)) \
    tests/tools/tp_stub_worker.py "$ROOT" "$GLUE" > "$OUT/tp_native_$n.log" 2>&1
  if [ $? -eq 0 ]; then echo "TP_NATIVE_OK world=$n (ns_tp_* + parallel_context glue over RCCL, one GPU per rank)"; else echo "(b) ns_tp over RCCL, $n ranks: FAIL (see $OUT/tp_native_$n.log)"; tail -5 "$OUT/tp_native_$n.log"; rc=1; fi
done
# ---- (c) ----------------------------------------------------------------------------------------------------------------------
if [ "${NS_FC_SKIP_BENCH:-0}" != "1" ]; then
  for n in 1 2 4 8; do
    if [ "$NGPU" -lt "$n" ]; then echo "(c) bench --gpus $n: SKIP (needs $n GPUs)"; continue; fi
    if [ "$n" -eq 1 ]; then python bench.py --gpus 1 --steps 20 --warmup 5 --chain-only > "$OUT/bench_1.json" 2> "$OUT/bench_1.err"
    else run_ranks "$n" $((29640 + n)) bench.py --gpus "$n" --steps 20 --warmup 5 > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"; fi
    python - "$OUT/bench_$n.json" "$n" <<'P' || rc=1
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d["config"]
    print("(c) N=%s: %.1f tokens/s, %.4f ms/step, all_reduce_us=%s, comm_fraction=%s, all_reduce=%s" % (
        sys.argv[2], d["value"], d["ms_per_step"], c.get("all_reduce_us"), c.get("comm_fraction"), c.get("all_reduce")))
except Exception as e:  # noqa: BLE001
    print("(c) N=%s: no JSON line (%s)" % (sys.argv[2], e)); sys.exit(1)
P
  done
else
  echo "(c) bench: SKIP (NS_FC_SKIP_BENCH=1)"
fi
[ $rc -eq 0 ] && echo "TP_FIRST_CONTACT_DONE gpus=$NGPU" || echo "TP_FIRST_CONTACT_FAILED gpus=$NGPU"
exit $rc
