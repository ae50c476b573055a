#!/bin/bash
# Developer tool (GPU box): the product's k_stream over and over, every run its own process under a timeout, output streamed — 10 x the seven small check
# scenes (bitwise against the quad kernel), 10 x C3 and 10 x C2 at the configs' sizes. usage: tools/stream_soak.sh TAG
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-soak}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
log=$out/k_stream_repeat.log; : > $log
for i in $(seq 1 10); do
  echo "run $i small" | tee -a $log
  WF_SHAPE=3 timeout -k 10 60 python3 tools/shape_check.py check 2>&1 | tee -a $log | tail -1 || { echo "FAILED or TIMED OUT (run $i small)" | tee -a $log; exit 1; }
done
for i in $(seq 1 10); do
  echo "run $i C3" | tee -a $log
  timeout -k 10 90 python3 tools/shape_check.py curly-hair 1280 32 3 2>&1 | tee -a $log | tail -1 || { echo "FAILED or TIMED OUT (run $i C3)" | tee -a $log; exit 1; }
  echo "run $i C2" | tee -a $log
  timeout -k 10 90 python3 tools/shape_check.py straight-hair 720 64 3 2>&1 | tee -a $log | tail -1 || { echo "FAILED or TIMED OUT (run $i C2)" | tee -a $log; exit 1; }
done
echo "all 30 runs completed" | tee -a $log
