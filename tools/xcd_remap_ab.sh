#!/bin/bash
# A/B of the XCD renumbering of the streaming kernel's workgroups (MGX_XCD_REMAP = 0 none,
# 1 neighbours along the strips share an XCD, 2 neighbours along the chunks) on the default
# benchmark; two runs each.  Prints ms per cycle and the level-0 up+down launch time.
set -eo pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for r in 0 1 2; do
    MGX_XCD_REMAP=$r timeout -k 10 120 python bench.py --steps 40 --warmup 5 --cpu-baseline off > gpurun_out/ab_tmp.json
    python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_tmp.json')); print('remap %s  ms/cycle %.4f  updown %.4f ms  frac %.3f' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))" $r
  done
done
