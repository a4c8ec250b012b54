#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_moe.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_gemvs.py -m gpu -q -x -k "config4 or split_k or decomposition" 2>&1 | tail -2; done
