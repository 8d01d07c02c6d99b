#!/bin/bash
# Sweep of the launch-shape knobs of the level-0 launches (environment variables read by
# csrc/mgx_stream.hip) on the default benchmark, tuner off so that the capped shape is used.
set -eo pipefail
run() {
  env "$@" MGX_TUNE=0 timeout -k 10 120 python bench.py --steps 30 --warmup 3 --cpu-baseline off > gpurun_out/sweep_tmp.json
  python3 -c "import json,sys; d=json.load(open('gpurun_out/sweep_tmp.json')); print('%-60s ms/cycle %.4f  updown %.4f ms' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['roofline']['avg_launch_ms']))" "$@"
}
run MGX_CAP_ROUNDS=3
run MGX_CAP_ROUNDS=2
run MGX_CAP_ROUNDS=4
run MGX_CAP_ROUNDS=6
run MGX_CAP_WGS=3 MGX_CAP_LDS=49152
run MGX_CAP_WGS=1 MGX_CAP_LDS=65536 MGX_CAP_ROUNDS=6
run MGX_CAP_CJ=256
run MGX_CAP_CJ=128
run MGX_CAP_ROUNDS=3 MGX_CAP_LDS=40000
run MGX_CAP_WGS=4 MGX_CAP_LDS=36864
run MGX_CAP_WGS=4 MGX_CAP_LDS=36864 MGX_CAP_ROUNDS=2
run MGX_CAP_WGS=3 MGX_CAP_LDS=49152 MGX_CAP_ROUNDS=2
run MGX_CAP_ROUNDS=5
run MGX_XCD_REMAP=0
