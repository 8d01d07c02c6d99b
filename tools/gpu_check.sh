set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | grep -v "Direct solver" | tail -3
timeout -k 10 120 python bench.py --cpu-baseline off > gpurun_out/occ.json
python3 -c "import json; d=json.load(open('gpurun_out/occ.json')); print('100 steps', round(d['ms_per_step'],4), round(d['value']/1e9,2))"
timeout -k 10 120 python bench.py --cpu-baseline off --slab > gpurun_out/occ.json
python3 -c "import json; d=json.load(open('gpurun_out/occ.json')); print('100 steps slab', round(d['ms_per_step'],4), round(d['value']/1e9,2))"
