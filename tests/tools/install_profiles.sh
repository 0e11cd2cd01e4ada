#!/bin/bash
# Copy the summaries of a profile round (tests/tools/profile_round.sh <tag> -> gpurun_out/<tag>/) into profiles/<round>_*.
#   tests/tools/install_profiles.sh <tag> <round>        e.g.  install_profiles.sh r05b r05
tag=$1; rnd=$2
s=gpurun_out/$tag
for f in bench.json hbm_traffic.json instruction_counters.json fp64_counters.json scratch_counters.json wait_counters.json provenance.json \
         serial_kernel_stats.csv serial_kernel_stats.csv.provenance.json overlapped_kernel_stats.csv \
         overlapped_kernel_stats.csv.provenance.json big_kernel_stats.csv big_kernel_stats.csv.provenance.json \
         serial_timeline.txt overlapped_timeline.txt e2e_stream.txt pmc_fetch_size.csv pmc_write_size.csv \
         pmc_instruction_counters.csv pmc_fp64_counters.csv pmc_scratch_counters.csv pmc_wait_counters.csv; do
  [ -s $s/$f ] && cp $s/$f profiles/${rnd}_$f
done
[ -s $s/stage_timers.txt ] && tail -1 $s/stage_timers.txt > profiles/${rnd}_stage_timers.json
ls -la profiles/${rnd}_* | wc -l
