#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scripts/moe_bench.py 1 2>gpurun_out/r04s_moe.err | tee gpurun_out/r04s_moe_mixtral_m1.json; tail -2 gpurun_out/r04s_moe.err
timeout 300 python scripts/moe_bench.py 8 2>>gpurun_out/r04s_moe.err | tee gpurun_out/r04s_moe_mixtral_m8.json
timeout 900 python scripts/full_token_oracle.py 2048 2>gpurun_out/r04s_ft.err | tee gpurun_out/r04s_full_token_vs_fp64_model.json; tail -3 gpurun_out/r04s_ft.err
