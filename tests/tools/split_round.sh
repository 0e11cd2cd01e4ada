#!/bin/bash
# One GPU round for the split window search: a first smoke of the pipeline (bounded), the parity suite, steady-state
# periods of the split pipeline against PW_SPLIT=0.  usage: split_round.sh TAG [quick]
tag=$1
mkdir -p gpurun_out/$tag
timeout 120 python tests/tools/sets_sweep.py 64 3 0,50,50 > gpurun_out/$tag/first.txt 2>&1; echo "first rc=$?"; tail -3 gpurun_out/$tag/first.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/gputest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed\|Error\|error" gpurun_out/$tag/gputest.log | tail -8
for n in 1000 4000 250; do
  it=20; [ $n = 4000 ] && it=8
  timeout 200 python tests/tools/sets_sweep.py $n $it 0,50,50 2>&1 | grep sets | sed "s/^/split n=$n /" >> gpurun_out/$tag/sweep.txt
  PW_SPLIT=0 timeout 200 python tests/tools/sets_sweep.py $n $it 0,50,50 2>&1 | grep sets | sed "s/^/fused-windows n=$n /" >> gpurun_out/$tag/sweep.txt
done
cat gpurun_out/$tag/sweep.txt
