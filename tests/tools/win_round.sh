#!/bin/bash
# One round for the window search: parity, stage timers, periods, device-vs-host fuzz.  usage: win_round.sh TAG
o=gpurun_out/$1; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cliffs.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | grep -a "passed\|failed" | tail -2
timeout 200 python tests/tools/profile_stages.py 1000 2>&1 | tail -1 > $o/stages.json
python - <<P
import json
d = json.load(open("$o/stages.json"))
u = d["us_per_unit"]
print({k: u[k] for k in ("win.path", "win.rotate", "win.z.step", "win.brute", "win.nm", "windows(total)", "sampling", "eps", "dbscan", "average", "opt.step")})
P
for n in 1000 4000; do it=30; [ $n = 4000 ] && it=10; timeout 200 python tests/tools/sets_sweep.py $n $it 0,-1,-1 2>&1 | grep sets | sed "s/^/n=$n /"; done
timeout 600 python tests/tools/fuzz_device_vs_host.py 1500 ${2:-41} 2>&1 | tail -2 | head -1
