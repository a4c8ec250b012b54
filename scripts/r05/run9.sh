#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out/r05j
python scripts/r05/prefill_cold_probe.py > gpurun_out/r05j/cold_probe.json 2> gpurun_out/r05j/cold_probe.err
cat gpurun_out/r05j/cold_probe.json
tail -3 gpurun_out/r05j/cold_probe.err
