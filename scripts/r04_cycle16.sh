#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
bash scripts/pmc_traffic.sh r04_pmc > gpurun_out/r04_pmc.log 2>&1; tail -3 gpurun_out/r04_pmc.log
cp gpurun_out/r04_pmc_fetch_size.json profiles/r04_pmc_fetch_size.json 2>/dev/null
echo "PMC done after $(( $(date +%s) - T0 )) s"
timeout 600 python bench.py > gpurun_out/r04r_bench.json 2>gpurun_out/r04r_bench.err
echo "BENCH exit $? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r04r_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04r_bench.json'))
c=d['config']
print('value', d['value'], 'roofline', d['roofline'])
print('prefill', c.get('prefill_m2048_tflops'), c.get('prefill_m2048_tflops_int8w'), c.get('prefill_m2048_tflops_ref_int8_semantics'))
print(json.dumps(c.get('prefill_m2048_detail'), indent=1))
print('config4', c.get('config4_tokens_per_s'), 'config5', c.get('config5_rank_ms'), 'full_token', c.get('full_token_tokens_per_s'))
print('cpu', d.get('cpu_baseline'))
PY
cp gpurun_out/r04_pmc_fetch_size.json gpurun_out/r04r_pmc_fetch_size.json 2>/dev/null
