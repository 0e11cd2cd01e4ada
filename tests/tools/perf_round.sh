#!/bin/bash
# Steady-state periods and single-analysis latency of the default pipeline (GPU box).  usage: perf_round.sh TAG [quick]
tag=$1
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/gputest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed" gpurun_out/$tag/gputest.log | tail -3
for n in 1000 4000 250; do
  it=30; [ $n = 4000 ] && it=10
  timeout 200 python tests/tools/sets_sweep.py $n $it 0,50,50 2>&1 | grep sets | sed "s/^/n=$n /" >> gpurun_out/$tag/sweep.txt
done
PW_TAIL_GATE=0 PW_HEAD_GATE=0 PW_SETS_IN_FLIGHT=2 timeout 200 python tests/tools/sets_sweep.py 1000 10 2,0,0 2>&1 | grep sets | sed "s/^/serial n=1000 /" >> gpurun_out/$tag/sweep.txt
cat gpurun_out/$tag/sweep.txt
