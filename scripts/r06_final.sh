#!/bin/bash
# round-6 evidence cycle: the device route's timings and traces, counters (own --pmc passes), the bench line, rocprofv3 statistics of the same command,
# the GPU suite, smoke.  Usage (GPU box): bash scripts/r06_final.sh [tag]    results under gpurun_out/, the ones to judge are copied into profiles/ afterwards
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r06z}
mkdir -p gpurun_out
T0=$(date +%s)
OUT=gpurun_out/${TAG}_route_timings.txt
D="python scripts/dev_llama7b.py"
flt() { grep "route timing\|prompt_eval\|us_median" | cut -c1-520; }
$D device 4 2048 > /dev/null 2>&1   # builds /tmp/ns_llama7b_q.bin
{
echo "scripts/dev_llama7b.py device 64 2048 (Llama-2-7B-shaped synthetic Q4_0 g32, the reference's own -DNS_SYCL build as the caller, 64 new tokens), NS_ROUTE_TIMING=1, one box"
for i in 1 2; do
echo "---- default, prompt 8, run $i";  NS_ROUTE_TIMING=1 timeout 300 $D device 64 2048 2>&1 | flt
echo "---- default, prompt 1500 (NS_HARNESS_PROMPT_REPEAT=1: the prompt is evaluated a second time in its context), run $i"
NS_ROUTE_TIMING=1 NS_HARNESS_PROMPT_REPEAT=1 NS_DEV7B_PROMPT=1500 timeout 300 $D device 64 2048 2>&1 | flt
done
for cfg in "NS_ROUTE_QKV_ROPE=0" "NS_DEVICE_KV=f32" "NS_ROUTE_LINKS=0" "NS_MHA_INLAUNCH=0" "NS_ROUTE_LAZY_SYNC=0" "NS_ROUTE_SEG=200" "NS_DEVICE_REPLAY=0"; do
echo "---- $cfg, prompt 1500"; env $cfg NS_ROUTE_TIMING=1 NS_DEV7B_PROMPT=1500 timeout 300 $D device 64 2048 2>&1 | grep "route timing: 61\|us_median" | cut -c1-420
done
for cfg in "NS_ROUTE_PREFILL_FUSE=0" "NS_ROUTE_WINDOW=0"; do
echo "---- $cfg, prompt 1500 (prompt figures)"; env $cfg NS_HARNESS_PROMPT_REPEAT=1 NS_DEV7B_PROMPT=1500 timeout 300 $D device 8 2048 2>&1 | grep "prompt_eval" | cut -c1-300
done
} > $OUT 2>&1
echo "ROUTE after $(( $(date +%s) - T0 )) s"; grep -c . $OUT
# kernels of the 1500-token run: statistics, the prompt's timeline, one replayed token's timeline
rm -rf gpurun_out/${TAG}_rprof
NS_DEV7B_PROMPT=1500 timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/${TAG}_rprof -o $TAG -- $D device 64 2048 > gpurun_out/${TAG}_rprof.out 2>&1
find gpurun_out/${TAG}_rprof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_route_1500_kernel_stats.csv \;
find gpurun_out/${TAG}_rprof -name "*kernel_trace.csv" -exec cp {} /tmp/${TAG}_trace.csv \;
find gpurun_out/${TAG}_rprof -name "*memory_copy_trace.csv" -exec cp {} /tmp/${TAG}_copy.csv \;
python scripts/route_timeline.py /tmp/${TAG}_trace.csv /tmp/${TAG}_copy.csv > gpurun_out/${TAG}_route_timeline.txt 2>&1
rm -rf gpurun_out/${TAG}_rprof
echo "ROUTE PROFILE after $(( $(date +%s) - T0 )) s"; head -3 gpurun_out/${TAG}_route_timeline.txt | cut -c1-200
bash scripts/pmc_traffic.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/${TAG}_pmc.log
cp gpurun_out/${TAG}_pmc_fetch_size.json profiles/r06_pmc_fetch_size.json
echo "PMC after $(( $(date +%s) - T0 )) s"
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2>gpurun_out/${TAG}_bench.err
echo "BENCH exit $? after $(( $(date +%s) - T0 )) s"; cut -c1-500 gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o $TAG -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-route > gpurun_out/${TAG}_bench_under_rocprofv3.json 2>/dev/null
find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_bench_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_prof
head -8 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-160
echo "ROCPROF after $(( $(date +%s) - T0 )) s"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_full_gpu_suite_pytest.txt 2>&1; tail -3 gpurun_out/${TAG}_full_gpu_suite_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
echo "ALL after $(( $(date +%s) - T0 )) s"
