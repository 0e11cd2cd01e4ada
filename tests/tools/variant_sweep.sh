#!/bin/bash
# Steady-state periods of library variants (tests/tools/build_variant.sh): variant_sweep.sh TAG name...  ("base" = the product library)
tag=$1; shift
mkdir -p gpurun_out/$tag
for v in "$@"; do
  if [ $v = base ]; then unset PW_LIB; else export PW_LIB=$PWD/tests/tools/libpw_var_$v.so; fi
  for n in ${PW_SWEEP_SIZES:-1000 125 4000}; do
    echo "== $v n=$n" >> gpurun_out/$tag/variants.txt
    timeout 120 python tests/tools/sets_sweep.py $n 20 0,50,50 2>&1 | grep sets >> gpurun_out/$tag/variants.txt
  done
done
cat gpurun_out/$tag/variants.txt
