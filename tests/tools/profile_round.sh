#!/bin/bash
# All profile passes of a round on the GPU box; writes gpurun_out/<tag>/ (copy the summaries to profiles/).
#   tests/tools/profile_round.sh <tag>
# Passes (each its own rocprofv3 run; counters never together with tracing):
#   1. bench.py itself (the judged line)                                  -> bench.json
#   2. kernel trace + stats of that same bench command (overlapped mode)  -> overlapped_kernel_stats.csv
#   3. kernel trace + stats, ONE analysis at a time (PW_TAIL_GATE=0 PW_HEAD_GATE=0, two sets): every
#      kernel duration is inside its step                                  -> serial_kernel_stats.csv, serial_timeline.txt
#   4. kernel trace of four overlapped analyses                           -> overlapped_timeline.txt
#   5. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes)               -> hbm_traffic.json
#   6. --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU            -> instruction_counters.json
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PW_STREAM_PROBE=0     # (counter passes serialise kernels: keep the pipeline's launch shape anyway)
T=$R/tests/tools
python3 $R/bench.py --steps 100 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ov -o ov --output-format csv -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary > $O/ov.log 2>&1
cp $(find $O/ov -name "*kernel_stats.csv" | head -1) $O/overlapped_kernel_stats.csv
( export PW_TAIL_GATE=0 PW_HEAD_GATE=0 PW_SETS_IN_FLIGHT=2
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/se -o se --output-format csv -- python3 $T/timeline.py 1000 20 > $O/se.log 2>&1 )
cp $(find $O/se -name "*kernel_stats.csv" | head -1) $O/serial_kernel_stats.csv
python3 $T/timeline_report.py $(find $O/se -name "*kernel_trace.csv" | head -1) | tail -40 > $O/serial_timeline.txt
timeout 300 rocprofv3 --kernel-trace -d $O/tl -o tl --output-format csv -- python3 $T/timeline.py 1000 6 > $O/tl.log 2>&1
python3 $T/timeline_report.py $(find $O/tl -name "*kernel_trace.csv" | head -1) | tail -60 > $O/overlapped_timeline.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf --output-format csv -- python3 $T/run_stage.py 15 1000 4 > $O/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw --output-format csv -- python3 $T/run_stage.py 15 1000 4 > $O/pw.log 2>&1
python3 $T/traffic_summary.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) 1000 $O/hbm_traffic.json > /dev/null
cp $(find $O/pf -name "*counter_collection.csv" | head -1) $O/pmc_fetch_size.csv
cp $(find $O/pw -name "*counter_collection.csv" | head -1) $O/pmc_write_size.csv
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $O/pv -o pv --output-format csv -- python3 $T/run_stage.py 15 1000 4 > $O/pv.log 2>&1
cp $(find $O/pv -name "*counter_collection.csv" | head -1) $O/pmc_instruction_counters.csv
python3 $T/counter_summary.py $O/pmc_instruction_counters.csv 1000 $O/instruction_counters.json > /dev/null
# 6b. what the vector ALUs execute in double precision (round-5 review, item 2): FMA / MUL / ADD / TRANS F64 wave instructions
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/p6 -o p6 --output-format csv -- python3 $T/run_stage.py 15 1000 4 > $O/p6.log 2>&1
cp $(find $O/p6 -name "*counter_collection.csv" | head -1) $O/pmc_fp64_counters.csv
python3 $T/fp64_summary.py $O/pmc_fp64_counters.csv 1000 $O/fp64_counters.json > /dev/null
# 7. what the scratch frames cost: scratch / flat instruction counts and the cycles waves spend waiting, per launch
#    (round-4 review item 4: "a counter experiment that prices it")
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $O/ps -o ps --output-format csv -- python3 $T/run_stage.py 15 1000 4 > $O/ps.log 2>&1
cp $(find $O/ps -name "*counter_collection.csv" | head -1) $O/pmc_scratch_counters.csv
python3 $T/counter_summary.py $O/pmc_scratch_counters.csv 1000 $O/scratch_counters.json > /dev/null
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pq -o pq --output-format csv -- python3 $T/run_stage.py 15 1000 4 > $O/pq.log 2>&1
cp $(find $O/pq -name "*counter_collection.csv" | head -1) $O/pmc_wait_counters.csv
python3 $T/counter_summary.py $O/pmc_wait_counters.csv 1000 $O/wait_counters.json > /dev/null
# 8. a molecule beyond LDS (pw_analyse_big_kernel): the 2865-atom shell of the capacity fixtures
timeout 300 rocprofv3 --kernel-trace --stats -d $O/bg -o bg --output-format csv -- python3 $T/big_unit_probe.py > $O/bg.log 2>&1
cp $(find $O/bg -name "*kernel_stats.csv" | head -1) $O/big_kernel_stats.csv
# 9. HISTORY file -> records: streamed against one piece
timeout 300 python3 $T/e2e_stream.py > $O/e2e_stream.txt 2>&1
rm -rf $O/ov $O/se $O/tl $O/pf $O/pw $O/pv $O/p6 $O/ps $O/pq $O/bg
ls -la $O
# 10. where these numbers come from: commit (.pw_head, written by `provenance.py stamp` before gpurun), date, hash of csrc/*
python3 $T/provenance.py annotate $O/hbm_traffic.json $O/instruction_counters.json $O/fp64_counters.json $O/scratch_counters.json $O/wait_counters.json $O/serial_kernel_stats.csv $O/overlapped_kernel_stats.csv $O/big_kernel_stats.csv > $O/provenance.json
