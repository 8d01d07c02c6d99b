#!/bin/bash
# Per-launch times of the small levels under different settings of the grouped launches
# (csrc/mgx_mid.hip).  Each line of the settings file (or the built-in list) is a set of
# VAR=value words; the benchmark runs under the kernel trace once per line at --size $SIZE.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/mid_exp.sh 4097'
set -eo pipefail
SIZE=${1:-4097}
LIST=${2:-}
OUT=gpurun_out/mid_exp
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$LIST" ]; then
    LIST=$OUT/settings.txt
    cat > $LIST <<'EOL'
MGX_NO_MID=1
MGX_MID=default
EOL
fi
n=0
while read -r line; do
    [ -z "$line" ] && continue
    n=$((n + 1))
    D=$OUT/run$n
    rm -rf $D
    # the settings are plain VAR=value words for this one command
    ( export $line; timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 bench.py --gpus 1 --size $SIZE --steps 20 --warmup 5 --cpu-baseline off > $D.json 2> $D.err )
    F=$(find $D -name '*kernel_trace.csv' | sed -n 1p)
    echo "=== $line  ($(python3 -c "import json; print(round(json.load(open('$D.json'))['ms_per_step'], 4))") ms/cycle)"
    python3 profiles/per_launch.py $F | awk '$NF == "us" && $(NF-1) < 60'
    python3 profiles/per_launch.py $F | awk '$NF == "us" && $(NF-1) < 60 {s += $(NF-1) * ($(NF-3) >= 20)} END {print "sum of launches under 60 us, once per cycle:", s, "us"}'
    rm -rf $D
done < $LIST
