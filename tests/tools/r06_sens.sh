#!/bin/bash
tag=$1; o=gpurun_out/$tag; mkdir -p $o
export PW_LIB=$PWD/tests/tools/libpw_var_sens.so
run() { timeout 100 python tests/tools/sets_sweep.py $1 $2 $3 2>&1 | grep -a "sets" | sed "s/^/$4 n=$1 /" >> $o/sweep.txt; }
for rep in 1 2; do
PW_CHAIN_PREP=0 run 1000 30 3,70,70 "prep off           "
PW_CHAIN_PREP=1 run 1000 30 3,70,70 "prep on            "
PW_CHAIN_PREP=1 PW_EXP_PREP_MODE=1 run 1000 30 3,70,70 "chains pay only    "
PW_CHAIN_PREP=1 PW_EXP_PREP_MODE=2 run 1000 30 3,70,70 "windows gain only  "
done
PW_CHAIN_PREP=0 run 4000 10 2,70,70 "prep off           "
PW_CHAIN_PREP=1 PW_EXP_PREP_MODE=1 run 4000 10 2,70,70 "chains pay only    "
PW_CHAIN_PREP=1 PW_EXP_PREP_MODE=2 run 4000 10 2,70,70 "windows gain only  "
cat $o/sweep.txt
