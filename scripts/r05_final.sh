#!/bin/bash
# round-5 evidence cycle: counters (own --pmc passes), the bench line, rocprofv3 statistics of the same command, the GPU suite, smoke.
# Usage (GPU box): bash scripts/r05_final.sh [tag]      results under gpurun_out/, the ones to judge are copied into profiles/ afterwards
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r05z}
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
# the prefill GEMM's SQ counters (own pass)
bash scripts/pmc_gemm.sh > gpurun_out/${TAG}_pmc_gemm3_sq_counters.txt 2>&1; tail -30 gpurun_out/${TAG}_pmc_gemm3_sq_counters.txt | cut -c1-200
echo "PMC GEMM after $(( $(date +%s) - T0 )) s"
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_full_gpu_suite_pytest.txt 2>&1; tail -3 gpurun_out/${TAG}_full_gpu_suite_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
echo "ALL after $(( $(date +%s) - T0 )) s"
