mkdir -p gpurun_out/r06m
for rep in 1 2; do for ct in 320 328 334 340 352; do echo -n "C teams $ct: "; PW_C_TEAMS=$ct timeout 100 python3 tests/tools/sets_sweep.py 1000 60 4,70,70 2>&1 | grep ms/step | cut -c1-80; done; done > gpurun_out/r06m/cteams_fine.txt 2>&1
cat gpurun_out/r06m/cteams_fine.txt
