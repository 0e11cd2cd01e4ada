#!/bin/bash
# Instruction-cache and wait counters of the three launches (one analysis at a time), separate --pmc passes
# (PW_PMC_SETS="A B;C D" replaces the default counter sets).
#   tests/tools/icache_round.sh <tag>   -> gpurun_out/<tag>/pmc_icache*.csv
tag=${1:-ic}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PW_STREAM_PROBE=0     # (counter passes serialise kernels: keep the pipeline's launch shape anyway)
T=$R/tests/tools
i=0
SETS=("SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS")
if [ -n "$PW_PMC_SETS" ]; then IFS=';' read -ra SETS <<< "$PW_PMC_SETS"; fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/p$i -o p$i --output-format csv -- python3 $T/run_stage.py 15 1000 4 > $O/p$i.log 2>&1
  cp $(find $O/p$i -name "*counter_collection.csv" | head -1) $O/pmc_icache_$i.csv
  rm -rf $O/p$i
done
python3 - $O <<'PY'
import csv, sys, glob, re, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/pmc_icache_*.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"pw_analyse_kernel<(\d+), (\d+)u?>", r["Kernel_Name"])
        if m: acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(sys.argv[1] + "/icache_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
