#!/bin/bash
# window teams that exclude each other from a CU by their LDS request (PW_C_LDS_KB), head gate nearly open
tag=$1; o=gpurun_out/$tag; mkdir -p $o
run() { timeout 100 python tests/tools/sets_sweep.py $1 $2 $3 2>&1 | grep -a "sets" | sed "s/^/$4 n=$1 /" >> $o/sweep.txt; }
for prep in 0 1; do
  export PW_CHAIN_PREP=$prep
  run 1000 30 "3,70,70" "prep=$prep base"
  for kb in 80 96; do
    export PW_C_LDS_KB=$kb
    run 1000 30 "3,70,70 3,70,1 3,70,30 3,50,1 4,70,1 4,50,1 3,30,1" "prep=$prep lds=$kb"
    unset PW_C_LDS_KB
  done
done
export PW_CHAIN_PREP=1 PW_C_LDS_KB=80
run 4000 10 "2,70,70 2,70,1 3,70,1" "prep=1 lds=80"
run 250 30 "4,50,50 4,50,1" "prep=1 lds=80"
cat $o/sweep.txt
