#!/bin/bash
# round-end GPU cycle under a tight budget: parity tests, weight-prefetch A/B sweep on the decode chain, the bench line,
# rocprofv3 kernel stats.  Every leg has its own timeout; results land in gpurun_out/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r01l}
mkdir -p gpurun_out
T0=$(date +%s)
timeout ${PYTEST_TIMEOUT:-330} python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_$TAG.log 2>&1
echo "PYTEST exit $? after $(( $(date +%s) - T0 )) s"; tail -4 gpurun_out/pytest_$TAG.log
: > gpurun_out/prefetch_sweep_$TAG.jsonl
for pf in "" "64:1.0" "256:1.0" "256:0.5" "128:1.0:fine" "256:1.0:fine" "512:1.0:fine"; do
  NS_BENCH_PREFETCH=$pf timeout 90 python bench.py --steps 200 --warmup 20 --chain-only 2>>gpurun_out/sweep_err.log \
    | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({'prefetch':'$pf','tok_s':d['value'],'ms':d['ms_per_step'],'chain_GBps':d['config']['chain_hbm_GBps'],'launch':d['config']['launch']}))" \
    | tee -a gpurun_out/prefetch_sweep_$TAG.jsonl
done
echo "SWEEP done after $(( $(date +%s) - T0 )) s"
timeout 200 python bench.py > gpurun_out/bench_$TAG.json 2>gpurun_out/bench_err.log
echo "BENCH exit $? after $(( $(date +%s) - T0 )) s"; cut -c1-400 gpurun_out/bench_$TAG.json
rm -rf gpurun_out/prof_$TAG
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench.json 2>/dev/null
echo "ROCPROF exit $? after $(( $(date +%s) - T0 )) s"
python scripts/trace_summary.py gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv smallm 2>&1 | tail -12
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete   # keep the merge under the 64 MiB cap
