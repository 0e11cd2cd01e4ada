#!/bin/bash
# every measurement of the working tree's library beside the round-5 kernels (tests/tools/libpw_var_base.so) on the SAME box
tag=$1; o=gpurun_out/$tag; mkdir -p $o
run() { timeout 100 python tests/tools/sets_sweep.py $1 $2 $3 2>&1 | grep -a "sets" | sed "s/^/$4 n=$1 /" >> $o/sweep.txt; }
base() { PW_LIB=$PWD/tests/tools/libpw_var_base.so run $1 $2 $3 "BASE(r5)          "; }
base 1000 30 3,70,70
for keep in 0 4; do
  PW_CHAIN_PREP=0 PW_TAIL_KEEP=$keep run 1000 30 "3,70,70 3,70,45 3,70,30" "prep=0 keep=$keep"
done
base 1000 30 3,70,70
base 4000 10 2,70,70
PW_CHAIN_PREP=0 PW_TAIL_KEEP=0 run 4000 10 "2,70,70 2,70,45" "prep=0 keep=0"
PW_CHAIN_PREP=0 PW_TAIL_KEEP=4 run 4000 10 "2,70,70 2,70,45" "prep=0 keep=4"
base 250 30 4,50,50
PW_CHAIN_PREP=0 PW_TAIL_KEEP=0 run 250 30 "4,50,50 4,50,30" "prep=0 keep=0"
PW_CHAIN_PREP=0 PW_TAIL_KEEP=4 run 250 30 "4,50,50 4,50,30" "prep=0 keep=4"
cat $o/sweep.txt
