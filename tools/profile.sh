#!/bin/bash
# Profiles of the default benchmark on the GPU box: kernel trace + stats, then the HBM
# traffic counters in two separate passes (never combined with a trace).  Outputs under
# gpurun_out/prof_<tag>/; summaries are made by profiles/summarize.py, profiles/per_launch.py
# and profiles/pmc_summary.py.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/profile.sh r02a'
set -eo pipefail
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline off > $OUT/bench_line_under_rocprof.json 2> $OUT/trace.err
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 profiles/summarize.py $F > $OUT/fine_level.txt
python3 profiles/per_launch.py $F > $OUT/per_launch.txt
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
cp $F $OUT/kernel_trace.csv
rm -rf $OUT/trace
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline off > /dev/null 2> $OUT/pmc_fetch.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline off > /dev/null 2> $OUT/pmc_write.err
python3 profiles/pmc_summary.py $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_write -name '*counter_collection.csv' | head -1) > $OUT/pmc_traffic.txt
head -3 $(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1) > $OUT/pmc_csv_head.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/fine_level.txt $OUT/pmc_traffic.txt
