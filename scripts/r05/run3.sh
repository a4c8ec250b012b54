#!/bin/bash
# round 5, GPU visit 3: f4 pair table by DMA (config 4) A-B + gemvs parity
set -x
mkdir -p gpurun_out/r05c
cd /root/repo
python -m pytest tests/test_gpu_gemvs.py tests/test_gpu_configs.py -x -q > gpurun_out/r05c/pytest.txt 2>&1
tail -3 gpurun_out/r05c/pytest.txt
for i in 1 2; do
NS_GVS_TABLE_DMA=0 python bench.py --secondary-only > gpurun_out/r05c/sec_valu_$i.json 2>> gpurun_out/r05c/err.txt
python bench.py --secondary-only > gpurun_out/r05c/sec_dma_$i.json 2>> gpurun_out/r05c/err.txt
done
python - <<'P'
import json
for n in ("sec_valu_1","sec_dma_1","sec_valu_2","sec_dma_2"):
    d=json.loads(open('gpurun_out/r05c/%s.json'%n).read().strip().splitlines()[-1])
    print(n, {k:(v.get('us_per_layer'),v.get('ms_per_step'),v.get('frac_of_8TBps')) for k,v in d.items() if isinstance(v,dict)})
P
