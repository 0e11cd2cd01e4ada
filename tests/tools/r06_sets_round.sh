#!/bin/bash
# Round 6: more analyses in flight with FEWER window teams each (so that every dispatch of every set can become resident),
# and two-wave window teams.  Every configuration has its own short time limit.  usage: r06_sets_round.sh TAG
tag=$1; o=gpurun_out/$tag; mkdir -p $o
export GPU_MAX_HW_QUEUES=32
run() { # n iters combo  (environment: PW_C_TEAMS, PW_B_TEAMS ...)
  timeout 40 python tests/tools/sets_sweep.py $1 $2 $3 2>&1 | grep -a "sets\|pywindow" | sed "s/^/$4 n=$1 /" >> $o/sweep.txt
}
run 125 40 4,50,50 base
for ct in 32 48 64; do
  PW_C_TEAMS=$ct PW_B_TEAMS=24 run 125 40 8,50,50 "C=$ct,B=24"
  PW_C_TEAMS=$ct PW_B_TEAMS=24 run 125 40 6,50,50 "C=$ct,B=24"
done
PW_C_TEAMS=32 PW_B_TEAMS=12 run 125 40 8,30,30 "C=32,B=12"
PW_C_TEAMS=64 PW_B_TEAMS=24 run 250 40 8,50,50 "C=64,B=24"
PW_C_TEAMS=64 PW_B_TEAMS=24 run 250 40 6,50,50 "C=64,B=24"
PW_C_TEAMS=96 PW_B_TEAMS=32 run 500 30 6,50,50 "C=96,B=32"
PW_C_TEAMS=128 PW_B_TEAMS=48 run 500 30 4,50,50 "C=128,B=48"
cat $o/sweep.txt
export PW_LIB=$PWD/tests/tools/libpw_var_nw2.so
for teams in 0 384; do
  export PW_C_WAVES=2; [ $teams != 0 ] && export PW_C_TEAMS=$teams
  timeout 60 python tests/tools/sets_sweep.py 1000 30 3,70,70 2>&1 | grep -a "sets\|pywindow" | sed "s/^/nw2 teams=$teams n=1000 /" >> $o/nw2.txt
  timeout 60 python tests/tools/sets_sweep.py 4000 10 2,70,70 2>&1 | grep -a "sets\|pywindow" | sed "s/^/nw2 teams=$teams n=4000 /" >> $o/nw2.txt
done
unset PW_C_WAVES PW_C_TEAMS
timeout 60 python tests/tools/sets_sweep.py 1000 30 3,70,70 2>&1 | grep -a "sets\|pywindow" | sed "s/^/nw4(variant lib) n=1000 /" >> $o/nw2.txt
cat $o/nw2.txt
