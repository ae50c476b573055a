#!/bin/bash
# Round 5: variants of k_stream's step (tools/_ab/libyhair_<name>.so) interleaved on one box: the seven check scenes against the quad kernel, then C2 / C3 / C4 timings.
# usage: r5_coop_ab.sh TAG "name1 name2 ..." [rounds]
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-coop}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
variants=${2:-"coop own"}; rounds=${3:-3}
for v in $variants; do
  echo "--- $v: k_stream (shape 3) against the quad kernel on the seven check scenes" | tee -a $out/ab.txt
  YHAIR_LIB=tools/_ab/libyhair_$v.so WF_SHAPE=3 timeout -k 10 300 python3 tools/shape_check.py check 2>&1 | grep -c "images equal True  rng equal True" | tee -a $out/ab.txt
done
for r in $(seq 1 $rounds); do
  for cfg in "straight-hair 720 192 3" "curly-hair 1280 64 3" "hair-curls 1280 64 3"; do
    for v in $variants; do
      printf "%s r%s: " $v $r | tee -a $out/ab.txt
      YHAIR_LIB=tools/_ab/libyhair_$v.so timeout -k 10 300 python3 tools/shape_check.py $cfg 2>&1 | grep Msamples | tail -1 | tee -a $out/ab.txt || exit 1
    done
  done
done
