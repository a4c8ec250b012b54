#!/bin/bash
# round-end GPU cycle under a tight budget: parity tests, weight-prefetch A/B sweep on the decode chain, the bench line,
# rocprofv3 kernel stats.  Every leg has its own timeout; results land in gpurun_out/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r01l}
mkdir -p gpurun_out
T0=$(date +%s)
timeout ${PYTEST_TIMEOUT:-330} python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_$TAG.log 2>&1
echo "PYTEST exit $? after $(( $(date +%s) - T0 )) s"; tail -4 gpurun_out/pytest_$TAG.log
if [ -n "$PREFETCH_SWEEP" ]; then
: > gpurun_out/prefetch_sweep_$TAG.jsonl
for pf in "" "64:1.0" "256:1.0" "256:0.5" "128:1.0:fine" "256:1.0:fine" "512:1.0:fine"; do
  NS_BENCH_PREFETCH=$pf timeout 90 python bench.py --steps 200 --warmup 20 --chain-only 2>>gpurun_out/sweep_err.log \
    | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({'prefetch':'$pf','tok_s':d['value'],'ms':d['ms_per_step'],'chain_GBps':d['config']['chain_hbm_GBps'],'launch':d['config']['launch']}))" \
    | tee -a gpurun_out/prefetch_sweep_$TAG.jsonl
done
fi
# peer-memory all-reduce latency with the ranks sharing this GPU (flags + fences + payload; no xGMI hop)
for ws in 2 4; do
  NS_P2P_LATENCY=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ws \
    --master-addr 127.0.0.1 --master-port $((29600+ws)) tests/p2p_worker.py 2>gpurun_out/p2p_err_$ws.log | grep -E "P2P_" | tee -a gpurun_out/p2p_latency_$TAG.txt
done
# the TP bench path with both ranks on this GPU (gloo process group, INVALID as a number): peer-memory vs process-group all-reduce
for p2p in 1 0; do
  NS_P2P=$p2p NS_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port $((29610+p2p)) bench.py --gpus 2 --steps 50 --warmup 5 2>gpurun_out/tp2_err_$p2p.log \
    | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({'tp2_on_one_gpu_p2p':$p2p,'tok_s':d['value'],'ms':d['ms_per_step'],'launch':d['config']['launch'],'all_reduce':d['config']['all_reduce']}))" | tee -a gpurun_out/tp2_one_gpu_$TAG.jsonl
done
echo "SWEEP done after $(( $(date +%s) - T0 )) s"
timeout 200 python bench.py > gpurun_out/bench_$TAG.json 2>gpurun_out/bench_err.log
echo "BENCH exit $? after $(( $(date +%s) - T0 )) s"; cut -c1-400 gpurun_out/bench_$TAG.json
rm -rf gpurun_out/prof_$TAG
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench.json 2>/dev/null
echo "ROCPROF exit $? after $(( $(date +%s) - T0 )) s"
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
python scripts/trace_summary.py gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv smallm 2>&1 | tail -12
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete   # keep the merge under the 64 MiB cap
