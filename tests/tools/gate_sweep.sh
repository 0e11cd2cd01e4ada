#!/bin/bash
# sweep of the two overlap gates (GPU box): prints ms_per_step per (tail, head)
for t in ${TAILS:-85 92 97 99}; do for h in ${HEADS:-85 92 97 99}; do
  v=$(PW_TAIL_GATE=$t PW_HEAD_GATE=$h timeout 200 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  echo "tail $t head $h ms $v"
done; done
