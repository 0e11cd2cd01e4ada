#!/bin/bash
# Host-side round on the GPU box: the multi-rank rehearsals, the API tests, bench.py with its end-to-end figures.  usage: host_round.sh TAG
tag=$1
o=gpurun_out/$tag
mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_api.py tests/test_zz_no_retries.py -m gpu -x -q > $o/gputest.log 2>&1
grep -a "passed\|failed\|Error" $o/gputest.log | tail -5
timeout 600 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err
python - <<P
import json
d = json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
e = d["secondary"]["e2e_history_to_records"]
print(d["value"], d["ms_per_step"], {k: e.get(k) for k in ("ms", "open_index_ms", "open_plus_analysis_ms", "cold_first_call_ms")})
print("retries", d["config"].get("retries"), "stale", d["roofline"].get("profile_inputs"))
P
