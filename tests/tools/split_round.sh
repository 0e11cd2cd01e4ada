#!/bin/bash
# One GPU round: the parity suite, then steady-state periods of the default pipeline and of the split window search.
# usage: split_round.sh TAG
tag=$1
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/gputest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed" gpurun_out/$tag/gputest.log | tail -3
for n in 1000 4000 250; do
  it=20; [ $n = 4000 ] && it=8
  timeout 200 python tests/tools/sets_sweep.py $n $it 0,50,50 2>&1 | grep sets | sed "s/^/default n=$n /" >> gpurun_out/$tag/sweep.txt
done
PW_SPLIT=1 timeout 200 python tests/tools/sets_sweep.py 1000 10 0,50,50 2>&1 | grep sets | sed "s/^/PW_SPLIT=1 n=1000 /" >> gpurun_out/$tag/sweep.txt
cat gpurun_out/$tag/sweep.txt
