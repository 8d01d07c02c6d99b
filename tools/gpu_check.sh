set -e
timeout -k 10 900 python -u -m pytest tests/test_gpu_slab.py -x -q --timeout 400 --timeout-method thread 2>&1 | grep -v "Direct solver" | tail -15
