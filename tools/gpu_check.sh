set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/kt
MGX_TUNE=0 timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline off > gpurun_out/pmc_f.log 2>&1
MGX_TUNE=0 timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline off > gpurun_out/pmc_w.log 2>&1
MGX_TUNE=0 timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt --output-format csv -- python3 bench.py --steps 20 --warmup 3 --cpu-baseline off > gpurun_out/kt.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json
python3 -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['ms_per_step'], d['value']/1e9, d['roofline']['avg_launch_ms'])"
