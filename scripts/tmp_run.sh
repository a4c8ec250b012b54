cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
timeout 200 python scripts/gemm_shapes_bench.py 2048 2>/dev/null | tail -1
timeout 200 python scripts/gemm_shapes_bench.py 2048 int8 2>/dev/null | tail -1
timeout 300 python scripts/prefill_bench.py 2048 512 2>/dev/null | tail -1
