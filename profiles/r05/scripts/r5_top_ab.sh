#!/bin/bash
# Round 5: the top of the traversal stack in a register (dev_trace.h) against reading it from LDS at the pop, interleaved on one box.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-topab}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for v in top notop; do
  echo "--- $v: scenes of the small check set whose images and RNG states equal the quad kernel's, for shapes 1 5 4 8 (7 each)" | tee -a $out/ab.txt
  for s in 1 5 4 8; do YHAIR_LIB=tools/_ab/libyhair_$v.so WF_SHAPE=$s timeout -k 10 120 python3 tools/shape_check.py check 2>&1 | grep -c "images equal True  rng equal True" | tee -a $out/ab.txt; done
done
for r in 1 2 3; do
  for cfg in "sphere-hairblock 720 64 5" "sphere-hairblock 720 64 0" "hair-curls 1280 32 1" "straight-hair 720 64 1" "sphere-hairblock 720 64 4 4"; do
    for v in top notop; do
      printf "%s r%s: " $v $r | tee -a $out/ab.txt
      YHAIR_LIB=tools/_ab/libyhair_$v.so timeout -k 10 200 python3 tools/shape_check.py $cfg 2>&1 | grep Msamples | tail -1 | tee -a $out/ab.txt || exit 1
    done
  done
done
