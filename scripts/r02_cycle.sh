#!/bin/bash
# round-2 GPU cycle: bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE pass, smoke
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02m}
mkdir -p gpurun_out/$TAG
T0=$(date +%s)
timeout 400 python bench.py > gpurun_out/$TAG/bench.json 2>gpurun_out/$TAG/bench_err.log
echo "BENCH exit $? after $(( $(date +%s) - T0 )) s"; cut -c1-2500 gpurun_out/$TAG/bench.json; tail -3 gpurun_out/$TAG/bench_err.log
rm -rf gpurun_out/$TAG/prof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -o $TAG -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/prof_bench.json 2>/dev/null
echo "ROCPROF exit $? after $(( $(date +%s) - T0 )) s"
head -12 gpurun_out/$TAG/prof/*/${TAG}_kernel_stats.csv 2>/dev/null || find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head
find gpurun_out/$TAG/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/$TAG/kernel_stats.csv \;
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -size +20M -delete
if [ -z "$NO_PMC" ]; then
  scripts/pmc_traffic.sh $TAG/pmc 2>&1 | tail -12
  find gpurun_out/$TAG/pmc -name "*.csv" -size +20M -delete
fi
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
echo "DONE after $(( $(date +%s) - T0 )) s"
