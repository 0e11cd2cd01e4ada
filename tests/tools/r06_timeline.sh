#!/bin/bash
# a kernel-trace timeline of N back-to-back analyses (steady state in the middle)   usage: r06_timeline.sh TAG [units] [launches]
tag=$1; n=${2:-1000}; k=${3:-16}; R=$PWD; o=$R/gpurun_out/$tag; mkdir -p $o
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $o/tl -o tl --output-format csv -- python3 $R/tests/tools/timeline.py $n $k > $o/tl.log 2>&1
python3 $R/tests/tools/timeline_report.py $(find $o/tl -name "*kernel_trace.csv" | head -1) > $o/timeline_${n}x$k.txt
rm -rf $o/tl; tail -3 $o/tl.log; wc -l $o/timeline_${n}x$k.txt
