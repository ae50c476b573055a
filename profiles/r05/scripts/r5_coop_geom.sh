#!/bin/bash
# Round 5: k_stream's launch geometry on C3 after the cooperative leaves (YHAIR_ST_SLOTS = slots per wave, YHAIR_ST_WAVES = waves per CU)
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-coopgeom}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for r in 1 2; do
  for g in "192 16" "128 16" "256 16" "320 16" "192 12" "256 12"; do
    set -- $g
    printf "slots %s waves/CU %s r%s: " $1 $2 $r | tee -a $out/geom.txt
    YHAIR_ST_SLOTS=$1 YHAIR_ST_WAVES=$2 timeout -k 10 300 python3 tools/shape_check.py curly-hair 1280 64 3 2>&1 | grep Msamples | tail -1 | tee -a $out/geom.txt || exit 1
  done
done
