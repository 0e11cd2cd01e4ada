# usage: r06_ab2.sh OUT VARIANT...   (each variant = tests/tools/libpw_var_<name>.so; "product" = the built library)
out=$1; shift; mkdir -p $(dirname $out)
for rep in 1 2; do
for v in "$@"; do
  if [ $v = product ]; then unset PW_LIB; else export PW_LIB=$PWD/tests/tools/libpw_var_$v.so; fi
  for n in 1000 4000; do echo -n "$v n=$n "; timeout 200 python3 tests/tools/sets_sweep.py $n 40 0,70,70 2>&1 | grep "ms/step"; done
done; done > $out 2>&1
cat $out
