#!/bin/bash
o=gpurun_out/$1; mkdir -p $o
for n in 4000 8192 20000; do
  for ct in 256 320 512; do echo "n=$n PW_C_TEAMS=$ct" | tee -a $o/resweep2.txt; PW_C_TEAMS=$ct timeout 300 python tests/tools/sets_sweep.py $n 6 0,50,50 2,50,50 2>&1 | grep sets | tee -a $o/resweep2.txt; done
done
for bt in 80 96 112; do for g in 0,50,50 3,70,70 3,30,60; do echo "n=1000 PW_B_TEAMS=$bt" | tee -a $o/resweep2.txt; PW_B_TEAMS=$bt timeout 200 python tests/tools/sets_sweep.py 1000 30 $g 2>&1 | grep sets | tee -a $o/resweep2.txt; done; done
for bt in 96 128; do echo "n=4000 C=256 PW_B_TEAMS=$bt" | tee -a $o/resweep2.txt; PW_C_TEAMS=256 PW_B_TEAMS=$bt timeout 200 python tests/tools/sets_sweep.py 4000 8 0,50,50 2>&1 | grep sets | tee -a $o/resweep2.txt; done
for bt in 64 96 128; do echo "n=250 PW_B_TEAMS=$bt" | tee -a $o/resweep2.txt; PW_B_TEAMS=$bt timeout 200 python tests/tools/sets_sweep.py 250 40 4,50,50 4,30,30 2>&1 | grep sets | tee -a $o/resweep2.txt; done
