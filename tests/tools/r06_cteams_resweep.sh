mkdir -p gpurun_out/r06i
for ct in 256 288 304 320 336 352; do echo "C teams $ct"; PW_C_TEAMS=$ct timeout 100 python3 tests/tools/sets_sweep.py 1000 40 4,70,70 2>&1 | grep ms/step; PW_C_TEAMS=$ct timeout 100 python3 tests/tools/sets_sweep.py 4000 20 4,70,70 2>&1 | grep ms/step; done > gpurun_out/r06i/cteams.txt 2>&1
cat gpurun_out/r06i/cteams.txt
