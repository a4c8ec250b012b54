#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05s
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05s/pytest_full.txt 2>&1
tail -8 gpurun_out/r05s/pytest_full.txt
