mkdir -p gpurun_out/r02k
for v in base occ3 occ3c; do
  for ct in 0 512; do
    for n in 1000 4000; do
      if [ $v = base ]; then unset PW_LIB; else export PW_LIB=$PWD/tests/tools/libpw_var_$v.so; fi
      if [ $ct = 0 ]; then unset PW_C_TEAMS; else export PW_C_TEAMS=$ct; fi
      echo "== $v C_TEAMS=$ct n=$n" >> gpurun_out/r02k/occ.txt
      timeout 120 python tests/tools/sets_sweep.py $n 10 0,50,50 2>&1 | grep sets >> gpurun_out/r02k/occ.txt
    done
  done
done
cat gpurun_out/r02k/occ.txt
