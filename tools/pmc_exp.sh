set -eo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_exp
mkdir -p $OUT
i=0
for cfg in "MGX_CAP_ROUNDS=3" "MGX_CAP_CJ=600 MGX_CAP_ROUNDS=1" "MGX_CAP_CJ=1200 MGX_CAP_ROUNDS=1 MGX_CAP_MINBLK=128"; do
  i=$((i+1))
  export $cfg
  MGX_TUNE=0 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f$i -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline off > $OUT/line$i.json 2> $OUT/err$i.txt
  MGX_TUNE=0 timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w$i -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline off > /dev/null 2>> $OUT/err$i.txt
  echo "== $cfg"
  python3 profiles/pmc_summary.py $(find $OUT/f$i -name '*counter_collection.csv' | head -1) $(find $OUT/w$i -name '*counter_collection.csv' | head -1) > $OUT/sum$i.txt; sed -n 2,3p $OUT/sum$i.txt
  python3 -c "import json; d=json.load(open('$OUT/line$i.json')); print('updown ms', d['roofline']['avg_launch_ms'])"
  rm -rf $OUT/f$i $OUT/w$i
  unset MGX_CAP_ROUNDS MGX_CAP_CJ MGX_CAP_MINBLK
done
