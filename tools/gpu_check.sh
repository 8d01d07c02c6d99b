set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_coarse.py -x -q --timeout 120 --timeout-method thread 2>&1 | grep -v "Direct solver" | tail -25
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | grep -v "Direct solver" | tail -5
