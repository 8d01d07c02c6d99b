set -e
mkdir -p gpurun_out
b() { python3 -c "import json; d=json.load(open('gpurun_out/occ.json')); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"; }
timeout -k 10 120 python bench.py --steps 50 --cpu-baseline off > gpurun_out/occ.json; b "default"
for cfg in "8 60" "8 30" "2 60" "2 30" "1 60"; do
set -- $cfg
MGX_EXP_NWV=$1 MGX_EXP_MINSTRIPS=$2 timeout -k 10 120 python bench.py --steps 50 --cpu-baseline off > gpurun_out/occ.json; b "plain nwv $1 minstrips $2"
done
for mb in 512 2048; do
MGX_MIN_BLOCKS=$mb timeout -k 10 120 python bench.py --steps 50 --cpu-baseline off > gpurun_out/occ.json; b "min_blocks $mb"
done
