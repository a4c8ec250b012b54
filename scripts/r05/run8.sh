#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out/r05i
timeout 600 python -m pytest tests/test_gpu_int8_mode.py -x -q > gpurun_out/r05i/pytest2.txt 2>&1
tail -2 gpurun_out/r05i/pytest2.txt
for i in 1 2; do
NS_I8_INKERNEL=0 python scripts/r05/i8_decode_ab.py 2>/dev/null | tail -1
python scripts/r05/i8_decode_ab.py 2>/dev/null | tail -1
done
