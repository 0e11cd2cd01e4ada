#!/bin/bash
# Round 6, first GPU call: baselines on this box + two probes (stream concurrency beyond twelve queues; the FP64 counters
# rocprofv3 knows).   usage: r06_probe_round.sh TAG
tag=$1; o=gpurun_out/$tag; mkdir -p $o
( cd tests/tools/streamprobe; for q in 12 16 24 32 64; do GPU_MAX_HW_QUEUES=$q timeout 60 ./stream_probe 10 12 16 18 20 24 32; done ) > $o/stream_probe.txt 2>&1
cat $o/stream_probe.txt
rocprofv3 -L 2>/dev/null | grep -i -B1 -A3 "F64\|FLOPS" | head -150 > $o/counters_f64.txt
wc -l $o/counters_f64.txt
export GPU_MAX_HW_QUEUES=12
for n in 1000 125 250 500 4000; do
  it=30; [ $n = 4000 ] && it=10
  g=70; [ $n -le 600 ] && g=50
  timeout 200 python tests/tools/sets_sweep.py $n $it 0,$g,$g 2>&1 | grep sets | sed "s/^/n=$n /" >> $o/sweep.txt
done
cat $o/sweep.txt
timeout 200 python3 tests/tools/profile_stages.py 1000 30 2>&1 | tail -1 > $o/stage_timers_steady.json
timeout 200 python3 tests/tools/profile_stages.py 1000 2>&1 | tail -1 > $o/stage_timers_single.json
cat $o/stage_timers_steady.json
