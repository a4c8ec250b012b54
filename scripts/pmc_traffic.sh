#!/bin/bash
# HBM traffic of the decode kernels from the TCC counters (own pass: --pmc with kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-pmc}
rm -rf gpurun_out/$TAG
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG -o $TAG -- python bench.py --steps 3 --warmup 1 --chain-only > gpurun_out/${TAG}_bench.json 2>gpurun_out/${TAG}_err.log
ls gpurun_out/$TAG | head
python scripts/pmc_summary.py gpurun_out/$TAG/${TAG}_counter_collection.csv gpurun_out/${TAG}_fetch_size.json
# ... and of BASELINE config 4's launches (gemvs_kernel: Mistral-7B NF4 g128, 8 rows): weight bytes per layer from the bench's own count
rm -rf gpurun_out/${TAG}_c4
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_c4 -o ${TAG}_c4 -- python bench.py --secondary-only > gpurun_out/${TAG}_c4_bench.json 2>gpurun_out/${TAG}_c4_err.log
LB=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_c4_bench.json'))['config4']['weight_bytes_per_layer'])")
python scripts/pmc_summary.py gpurun_out/${TAG}_c4/${TAG}_c4_counter_collection.csv --config4 gpurun_out/${TAG}_config4_fetch_size.json $LB
