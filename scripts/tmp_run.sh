cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_decoder_layer.py -q -x 2>&1 | tail -4
timeout 300 python scripts/config_bench.py 2>/dev/null > gpurun_out/cfg.json; python -c "
import json; d=json.load(open('gpurun_out/cfg.json'))
for k in ('config4_mistral7b_nf4_g128_batch8',):
    v=d[k]; print(k, {a:b for a,b in v.items() if a!='per_shape'}); print('   ', {a:(b['us'], b.get('GBps')) for a,b in v['per_shape'].items()})
"
