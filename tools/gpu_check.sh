set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_fmg.py -x -q --timeout 300 --timeout-method thread -k 32769 2>&1 | tail -15
timeout -k 10 200 python bench.py --size 22529 --coarsest 11 --exact-coarse --steps 20 --cpu-baseline off > gpurun_out/b22529.json
python3 -c "import json; d=json.load(open('gpurun_out/b22529.json')); print(d['ms_per_step'], d['value']/1e9, d['config']['workload'], d['resnorm_first_last'])"
timeout -k 10 200 python bench.py --size 22529 --coarsest 11 --exact-coarse --steps 20 --cpu-baseline off --slab > gpurun_out/b22529s.json
python3 -c "import json; d=json.load(open('gpurun_out/b22529s.json')); print(d['ms_per_step'], d['value']/1e9, d['config']['decomposition'], d['resnorm_first_last'])"
timeout -k 10 200 python bench.py --size 32769 --steps 20 --cpu-baseline off > gpurun_out/b32769.json
python3 -c "import json; d=json.load(open('gpurun_out/b32769.json')); print(d['ms_per_step'], d['value']/1e9, d['resnorm_first_last'])"
