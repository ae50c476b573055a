#!/bin/bash
# Round 5, first GPU call: (1) the gate of VERDICT r04 item 1 — idle quad trips of the quad kernel on C1 / C2 / C3 / C4 from the
# instrumented build's counters; (2) VERDICT item 3 — the product's k_stream 30 x under a timeout, streamed, on the small scene and on C3.
cd $GRAFT_REPO_ROOT; out=gpurun_out/r5gate; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for cfg in "sphere-hairblock 720 0" "straight-hair 720 1" "curly-hair 1280 1" "hair-curls 1280 1"; do
  set -- $cfg
  timeout -k 10 300 python3 tools/chain_profile.py $1 $2 $3 2>&1 | grep -v "^\[yhair\]" | tee -a $out/quad_idle_trips.txt || exit 1
done
echo "--- k_stream repeat (hang reproduction) ---" | tee $out/k_stream_repeat.log
for i in $(seq 1 15); do
  echo "run $i small" | tee -a $out/k_stream_repeat.log
  WF_SHAPE=3 timeout -k 10 60 python3 tools/shape_check.py check 2>&1 | tee -a $out/k_stream_repeat.log | tail -1 || { echo "FAILED or TIMED OUT rc=$? (run $i small)" | tee -a $out/k_stream_repeat.log; exit 1; }
done
for i in $(seq 1 15); do
  echo "run $i C3" | tee -a $out/k_stream_repeat.log
  timeout -k 10 90 python3 tools/shape_check.py curly-hair 1280 32 3 2>&1 | tee -a $out/k_stream_repeat.log | tail -1 || { echo "FAILED or TIMED OUT rc=$? (run $i C3)" | tee -a $out/k_stream_repeat.log; exit 1; }
done
echo "all 30 runs completed" | tee -a $out/k_stream_repeat.log
