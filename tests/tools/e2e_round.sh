#!/bin/bash
# End to end (HISTORY file -> records): streamed against one-piece, medians (GPU box).  usage: e2e_round.sh TAG
tag=$1
mkdir -p gpurun_out/$tag
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "streamed or threads" > gpurun_out/$tag/gputest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed" gpurun_out/$tag/gputest.log | tail -3
timeout 300 python tests/tools/e2e_stream.py > gpurun_out/$tag/e2e.txt 2>&1
cat gpurun_out/$tag/e2e.txt
