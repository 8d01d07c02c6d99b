#!/bin/bash
# A/B of builds of libmgx on the default benchmark: tools/ab.sh <lib.so> [<lib.so> ...]
# (MGX_LIB selects the library).  Prints ms per cycle and the level-0 up+down launch time.
set -eo pipefail
for lib in "$@"; do
  for rep in 1 2; do
    MGX_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 40 --warmup 5 --cpu-baseline off > gpurun_out/ab_tmp.json
    python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_tmp.json')); print('%-50s ms/cycle %.4f  updown %.4f ms  frac %.3f' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))" $lib
  done
done
