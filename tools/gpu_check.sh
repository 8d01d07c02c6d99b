set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
timeout -k 10 120 python bench.py --steps 20 --cpu-baseline off > gpurun_out/bench_20.json
python -c "import json; print('20 steps', json.load(open('gpurun_out/bench_20.json'))['ms_per_step'])"
