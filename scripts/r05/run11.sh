#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05l
NS_ROUTE_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_route_replay.py -x -q > gpurun_out/r05l/pytest.txt 2>&1
tail -30 gpurun_out/r05l/pytest.txt | cut -c1-400
