#!/bin/bash
# round 6, GPU call 10: k_stream's pool size after the slot's sectors (host-side knob only: YHAIR_ST_SLOTS), C3 and C2, interleaved
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g10; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for r in 1 2; do
  for cfg in "curly-hair 1280 64" "straight-hair 720 64"; do
    for sl in default 128 192 256 320; do
      printf "slots %s r%s: " $sl $r | tee -a $out/slots.txt
      if [ $sl = default ]; then timeout -k 10 200 python3 tools/shape_check.py $cfg 3 2>&1 | grep Msamples | tail -1 | tee -a $out/slots.txt
      else YHAIR_ST_SLOTS=$sl timeout -k 10 200 python3 tools/shape_check.py $cfg 3 2>&1 | grep Msamples | tail -1 | tee -a $out/slots.txt; fi
    done
  done
done
