#!/bin/bash
# HBM traffic of the decode kernels from the TCC counters (own pass: --pmc with kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-pmc}
rm -rf gpurun_out/$TAG
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG -o $TAG -- python bench.py --steps 3 --warmup 1 --chain-only > gpurun_out/${TAG}_bench.json 2>gpurun_out/${TAG}_err.log
ls gpurun_out/$TAG | head
python scripts/pmc_summary.py gpurun_out/$TAG/${TAG}_counter_collection.csv gpurun_out/${TAG}_fetch_size.json
