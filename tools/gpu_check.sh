set -e
mkdir -p gpurun_out
b() { python3 -c "import json; d=json.load(open('gpurun_out/occ.json')); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"; }
for size in 32769 16385 8193 4097 2049; do
timeout -k 10 120 python bench.py --size $size --steps 50 --cpu-baseline off > gpurun_out/occ.json; b "$size default"
done
timeout -k 10 120 python bench.py --steps 50 --cpu-baseline off --slab > gpurun_out/occ.json; b "16385 slab"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | grep -v "Direct solver" | tail -4
