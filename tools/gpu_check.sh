set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | grep -v "Direct solver" | tail -4
b() { python3 -c "import json; d=json.load(open('gpurun_out/occ.json')); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['all_fine_level_kernels_ms'])"; }
timeout -k 10 120 python bench.py --steps 20 --cpu-baseline off > gpurun_out/occ.json; b "20 steps"
timeout -k 10 120 python bench.py --cpu-baseline off > gpurun_out/occ.json; b "100 steps"
