#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out/r05k
for i in 1 2; do
python bench.py --secondary-only > gpurun_out/r05k/sec_default_$i.json 2>> gpurun_out/r05k/err.txt
NS_GVS_TABLE=1 python bench.py --secondary-only > gpurun_out/r05k/sec_table_always_$i.json 2>> gpurun_out/r05k/err.txt
done
python - <<'P'
import json
for n in ("sec_default_1","sec_table_always_1","sec_default_2","sec_table_always_2"):
    d=json.loads(open('gpurun_out/r05k/%s.json'%n).read().strip().splitlines()[-1])
    print(n, {k:(v.get('us_per_layer'),v.get('ms_per_step'),v.get('frac_of_8TBps')) for k,v in d.items() if isinstance(v,dict)})
P
