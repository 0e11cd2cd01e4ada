mkdir -p gpurun_out/r06d
( for t in 40 50 60 70 80 90; do for h in 50 70 90; do echo "4,$t,$h"; done; done | xargs timeout 600 python3 tests/tools/sets_sweep.py 1000 40 ) > gpurun_out/r06d/gates.txt 2>&1
for ct in 288 320 352 384; do echo "C teams $ct"; PW_C_TEAMS=$ct timeout 100 python3 tests/tools/sets_sweep.py 1000 40 4,70,70 3,70,70; done > gpurun_out/r06d/cteams.txt 2>&1
tail -30 gpurun_out/r06d/gates.txt; cat gpurun_out/r06d/cteams.txt
