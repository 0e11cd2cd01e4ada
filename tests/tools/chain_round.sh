#!/bin/bash
# One measurement round for the optimiser chains (GPU box): parity first, then the chains-only launch, the sub-phase
# timers and the steady-state step.   usage: chain_round.sh TAG [full]
tag=$1
mkdir -p gpurun_out/$tag
o=gpurun_out/$tag
if [ "$2" = full ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > $o/gputest.log 2>&1; echo "pytest rc=$?"
else
  timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -m gpu -x -q -k "division or golden or parity or live or opt or lbfgsb or static or every_launch" > $o/gputest.log 2>&1; echo "pytest rc=$?"
fi
grep -a "passed\|failed\|error" $o/gputest.log | tail -3
# variant libraries (tests/tools/libpw_var_*.so, e.g. the forced-fallback build): the parity subset through each
for var in tests/tools/libpw_var_*.so; do
  [ -f "$var" ] || continue
  cp pywindow_amd/libpywindow_hip.so /tmp/pw_keep.so && cp "$var" pywindow_amd/libpywindow_hip.so
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_cliffs.py -m gpu -x -q > $o/gputest_$(basename $var .so).log 2>&1; echo "variant $(basename $var) rc=$?"
  grep -a "passed\|failed\|error" $o/gputest_$(basename $var .so).log | tail -2
  cp /tmp/pw_keep.so pywindow_amd/libpywindow_hip.so
done
timeout 200 python tests/tools/chains_only.py 1000 2>&1 | grep "product chains" | tee $o/chains_only.txt
timeout 300 python tests/tools/profile_chains.py 1000 > $o/fine.json 2> $o/fine.err; python - <<P
import json
d=json.load(open("$o/fine.json"))
print("fine kernel_ms", round(d["kernel_ms"],3), "nit", d["mean_nit"])
print(" coarse us/unit:", d["us_per_unit(100MHz)"])
print(" cycles/iteration:", d["cycles_per_iteration"])
P
timeout 200 python tests/tools/sets_sweep.py 1000 30 0,50,50 2>&1 | grep sets | sed "s/^/n=1000 /" | tee $o/sweep.txt
PW_TAIL_GATE=0 PW_HEAD_GATE=0 PW_SETS_IN_FLIGHT=2 timeout 200 python tests/tools/sets_sweep.py 1000 10 2,0,0 2>&1 | grep sets | sed "s/^/serial n=1000 /" | tee -a $o/sweep.txt
