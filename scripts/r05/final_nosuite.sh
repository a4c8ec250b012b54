#!/bin/bash
# the evidence cycle of scripts/r05_final.sh without the GPU suite / smoke (run separately: profiles/r05y_full_gpu_suite_pytest.txt)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r05y}
mkdir -p gpurun_out
T0=$(date +%s)
bash scripts/pmc_traffic.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1; tail -2 gpurun_out/${TAG}_pmc.log
cp gpurun_out/${TAG}_pmc_fetch_size.json profiles/r05_pmc_fetch_size.json
echo "PMC after $(( $(date +%s) - T0 )) s"
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2>gpurun_out/${TAG}_bench.err
echo "BENCH exit $? after $(( $(date +%s) - T0 )) s"; cut -c1-700 gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/${TAG}_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o $TAG -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_under_rocprofv3.json 2>/dev/null
find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_bench_kernel_stats.csv \;
find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete
head -12 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-160
echo "ROCPROF after $(( $(date +%s) - T0 )) s"
