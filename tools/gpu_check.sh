set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_dropin.py tests/test_cli.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -15
