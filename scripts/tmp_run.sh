cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_norm_link.py tests/test_gpu_attention.py tests/test_gpu_decoder_layer.py tests/test_gpu_parity.py -q -x 2>&1 | tail -5
timeout 120 scripts/ubench/chain_bench --modes 1 --reps 20 2>&1 | grep -i "chain\|tok" | head -3 | cut -c1-200
scripts/prof_full.sh r02p
