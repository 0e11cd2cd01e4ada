#!/bin/bash
# A/B of one knob on one box: parity subset, then periods with the knob off and on.   usage: r06_ab.sh TAG ENVNAME [full]
tag=$1; knob=$2; o=gpurun_out/$tag; mkdir -p $o
if [ "$3" = full ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $o/gputest.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/gputest.txt
else
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $o/gputest.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/gputest.txt
fi
for rep in 1 2; do
for v in 0 1; do
  for n in 1000 4000 250; do
    it=30; [ $n = 4000 ] && it=10
    g=70; [ $n -le 600 ] && g=50
    env $knob=$v timeout 100 python tests/tools/sets_sweep.py $n $it 0,$g,$g 2>&1 | grep -a "sets" | sed "s/^/$knob=$v n=$n /" >> $o/sweep.txt
  done
done
done
cat $o/sweep.txt
timeout 100 python3 tests/tools/profile_stages.py 1000 30 2>&1 | tail -1 > $o/stage_timers_steady.json; cat $o/stage_timers_steady.json
