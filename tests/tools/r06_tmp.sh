bash tests/tools/r06_timeline.sh r06x 125 16
PW_CONTEXT_TIMING=1 python tests/tools/first_call_time.py 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
