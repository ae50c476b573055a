#!/bin/bash
# Interleaved A/B of library variants (tools/_ab/libyhair_<name>.so) on forced launch shapes.
# usage: variants_ab.sh TAG "name1 name2" "scene res spp shape" ...
set -o pipefail  # (`|| exit 1` below tests the whole pipeline, not its last stage)
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
variants=$2; shift 2
cfgs=("$@")
for v in $variants; do
  echo "--- $v: scenes of the small check set whose images and RNG states equal the quad kernel's, for shapes 1 5 4 8 (7 each)" | tee -a $out/ab.txt
  for s in 1 5 4 8; do YHAIR_LIB=tools/_ab/libyhair_$v.so WF_SHAPE=$s timeout -k 10 120 python3 tools/shape_check.py check 2>&1 | grep -c "images equal True  rng equal True" | tee -a $out/ab.txt || { echo "$v shape $s: check FAILED or TIMED OUT" | tee -a $out/ab.txt; exit 1; }; done
done
for r in 1 2 3; do
  for cfg in "${cfgs[@]}"; do
    for v in $variants; do
      printf "%s r%s: " $v $r | tee -a $out/ab.txt
      YHAIR_LIB=tools/_ab/libyhair_$v.so timeout -k 10 200 python3 tools/shape_check.py $cfg 2>&1 | grep Msamples | tail -1 | tee -a $out/ab.txt || exit 1
    done
  done
done
