#!/bin/bash
# What a GPU session runs before a commit: the GPU parity suite, then the default benchmark.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_check.sh'
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | grep -v "Direct solver" | tail -4
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json
python3 -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print('ms/cycle', round(d['ms_per_step'],4), 'G DOF/s', round(d['value']/1e9,2))"
