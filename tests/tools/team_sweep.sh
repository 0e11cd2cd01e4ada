#!/bin/bash
# sweep of the persistent-team caps (GPU box): prints ms_per_step per (C teams, B teams)
for c in ${CT:-320 416 512}; do for b in ${BT:-256 512 1000}; do
  v=$(PW_C_TEAMS=$c PW_B_TEAMS=$b timeout 200 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  echo "C $c B $b ms $v"
done; done
