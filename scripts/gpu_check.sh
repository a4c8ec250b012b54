#!/bin/bash
# standard GPU cycle: parity tests, bench line, per-shape kernel durations (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-chk}
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>gpurun_out/bench_err.log > gpurun_out/bench_$TAG.json
python -c "
import json; d=json.load(open('gpurun_out/bench_$TAG.json')); r=d['roofline']
print('BENCH tok/s', d['value'], 'ms', d['ms_per_step'], 'chain GB/s', d['config']['chain_hbm_GBps'], '| gate/up', r['achieved'], 'GB/s', r['avg_launch_us'], 'us frac', r['frac'])"
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench.json 2>/dev/null
python scripts/trace_summary.py gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv smallm
