#!/bin/bash
# Everything a round's record needs, on one box: the whole GPU suite, the device-vs-host fuzz, the soak, the profile
# passes, the stage and chain timers.   usage: final_round.sh TAG     (writes gpurun_out/TAG/)
tag=$1; o=gpurun_out/$tag; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q > $o/gputest.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/gputest.txt
for seed in 61 62; do timeout 600 python tests/tools/fuzz_device_vs_host.py 3000 $seed 2>&1 | tail -2 | head -1; done | tee $o/fuzz.txt
PW_FUZZ_MAX_ATOMS=420 timeout 600 python tests/tools/fuzz_device_vs_host.py 1000 63 2>&1 | tail -2 | head -1 | tee -a $o/fuzz.txt
timeout 600 python tests/tools/soak_growth.py 300 5 2>&1 | tail -1 | tee $o/soak.txt
bash tests/tools/profile_round.sh $tag > $o/profile_round.log 2>&1
timeout 200 python3 tests/tools/profile_stages.py 1000 2>&1 | tail -1 > $o/stage_timers.txt
timeout 300 python3 tests/tools/profile_chains.py 1000 2>/dev/null > $o/chain_subphase_timers.json
timeout 200 python3 tests/tools/chains_only.py 1000 2>&1 | grep "product chains" | tee $o/chains_only.txt
