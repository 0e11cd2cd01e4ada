#!/bin/bash
# One GPU round of the inner loop: parity tests, in-kernel stage timers, steady-state periods.  usage: quick_round.sh TAG
tag=$1
mkdir -p gpurun_out/$tag
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/gputest.log 2>&1; grep -a "passed\|failed" gpurun_out/$tag/gputest.log
timeout 120 python tests/tools/profile_stages.py 1000 > gpurun_out/$tag/stages.json 2>&1
timeout 200 python tests/tools/sets_sweep.py 1000 20 0,50,50 2>&1 | grep sets > gpurun_out/$tag/sweep.txt
timeout 100 python tests/tools/sets_sweep.py 125 40 0,50,50 2>&1 | grep sets >> gpurun_out/$tag/sweep.txt
timeout 200 python tests/tools/sets_sweep.py 4000 10 0,50,50 2>&1 | grep sets >> gpurun_out/$tag/sweep.txt
cat gpurun_out/$tag/sweep.txt
