#!/bin/bash
# round 4, first GPU cycle: parity of the new small-batch kernel, then the knob sweep, then config_bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T0=$(date +%s)
timeout 400 python -m pytest tests/test_gpu_gemvs.py -m gpu -q -x --durations=5 > gpurun_out/r04a_pytest_gemvs.log 2>&1
echo "PYTEST gemvs exit $? after $(( $(date +%s) - T0 )) s"; tail -15 gpurun_out/r04a_pytest_gemvs.log
timeout 420 python scripts/gvs_sweep.py all > gpurun_out/r04a_gvs_sweep.txt 2>gpurun_out/r04a_gvs_sweep.err
echo "SWEEP exit $? after $(( $(date +%s) - T0 )) s"; grep -v "^{" gpurun_out/r04a_gvs_sweep.txt; tail -3 gpurun_out/r04a_gvs_sweep.err
timeout 300 python scripts/config_bench.py > gpurun_out/r04a_config_bench.json 2>gpurun_out/r04a_config_bench.err
echo "CONFIG exit $? after $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_config_bench.json'))
for k in ('config4_mistral7b_nf4_g128_batch8','config5_llama70b_q4_0_rank_of_tp8'):
    print(k, json.dumps(d[k]['graph_chain']), {n:v['us'] for n,v in d[k]['per_shape'].items()})
PY
