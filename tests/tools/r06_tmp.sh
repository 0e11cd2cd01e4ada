o=gpurun_out/r06k; mkdir -p $o
run() { timeout 100 python tests/tools/sets_sweep.py $1 $2 $3 2>&1 | grep -a "sets" | sed "s/^/$4 n=$1 /" >> $o/sweep.txt; }
PW_LIB=$PWD/tests/tools/libpw_var_base.so run 1000 30 3,70,70 "BASE(r5)    "
PW_CHAIN_PREP=0 PW_LIB=$PWD/tests/tools/libpw_var_claimfirst.so run 1000 30 3,70,70 "claim-first "
PW_CHAIN_PREP=0 PW_TAIL_KEEP=0 run 1000 30 3,70,70 "check-first "
PW_LIB=$PWD/tests/tools/libpw_var_base.so run 1000 30 3,70,70 "BASE(r5)    "
cat $o/sweep.txt
