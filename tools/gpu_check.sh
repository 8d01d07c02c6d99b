#!/bin/bash
# What a GPU session runs before a commit: the GPU parity suite, then the benchmark with the
# driver's arguments.  Every step has its own time limit and a failure ends the script.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_check.sh'
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 \
    || { grep -v "Direct solver" gpurun_out/gpu_tests.log | tail -40; exit 1; }
grep -v "Direct solver" gpurun_out/gpu_tests.log | tail -4
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_args.json
python3 -c "import json; d=json.load(open('gpurun_out/bench_driver_args.json')); print('ms/cycle', round(d['ms_per_step'],4), 'G DOF/s', round(d['value']/1e9,2), 'frac', round(d['roofline']['frac'],3))"
