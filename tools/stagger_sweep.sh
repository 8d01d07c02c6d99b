#!/bin/bash
# Does the placement of a level's three arrays relative to the memory's channel interleave
# matter?  MGX_STAGGER="a,b" starts f / the second iterate buffer a / b bytes into their
# allocations.  Prints ms per cycle and the level-0 up+down launch time, tuner off.
set -eo pipefail
mkdir -p gpurun_out
run() {
  MGX_STAGGER=$1 MGX_TUNE=0 timeout -k 10 120 python bench.py --steps 30 --warmup 3 --cpu-baseline off > gpurun_out/sweep_tmp.json
  python3 -c "import json,sys; d=json.load(open('gpurun_out/sweep_tmp.json')); print('%-24s ms/cycle %.4f  updown %.4f ms' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms']))" $1
}
for s in ${STAGGERS:-0,0 262144,524288 0,0 262144,524288 0,0 262144,524288 16384,32768 131072,262144 524288,1048576 393216,786432}; do run $s; done
