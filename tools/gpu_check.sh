set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_fmg.py -x -q --timeout 120 --timeout-method thread 2>&1 | grep -v "Direct solver" | tail -25
