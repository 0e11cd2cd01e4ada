#!/bin/bash
# Re-sweep of the pipeline's scheduling knobs after the kernels changed (GPU box).  usage: resweep.sh TAG
o=gpurun_out/$1; mkdir -p $o
for n in 1000 250 4000; do
  it=30; [ $n = 4000 ] && it=8
  echo "== n=$n gates/sets" | tee -a $o/resweep.txt
  timeout 600 python tests/tools/sets_sweep.py $n $it 0,50,50 2,50,50 3,50,50 4,50,50 3,30,30 3,70,70 3,30,60 3,60,30 3,80,85 4,30,30 4,70,70 2>&1 | grep sets | tee -a $o/resweep.txt
done
echo "== n=1000 window teams" | tee -a $o/resweep.txt
for ct in 128 192 256 384 512; do echo "PW_C_TEAMS=$ct" | tee -a $o/resweep.txt; PW_C_TEAMS=$ct timeout 200 python tests/tools/sets_sweep.py 1000 30 0,50,50 2>&1 | grep sets | tee -a $o/resweep.txt; done
echo "== n=1000 average teams" | tee -a $o/resweep.txt
for bt in 64 96 128 192 256; do echo "PW_B_TEAMS=$bt" | tee -a $o/resweep.txt; PW_B_TEAMS=$bt timeout 200 python tests/tools/sets_sweep.py 1000 30 0,50,50 2>&1 | grep sets | tee -a $o/resweep.txt; done
echo "== n=4000 window teams" | tee -a $o/resweep.txt
for ct in 256 384 512; do echo "PW_C_TEAMS=$ct" | tee -a $o/resweep.txt; PW_C_TEAMS=$ct timeout 200 python tests/tools/sets_sweep.py 4000 8 0,50,50 2>&1 | grep sets | tee -a $o/resweep.txt; done
