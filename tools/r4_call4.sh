#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4d; mkdir -p $out; export TMPDIR=/tmp
WF_SHAPE=3 python3 tools/shape_check.py check > $out/check.txt 2>&1; tail -8 $out/check.txt
for cfg in "straight-hair 720 64" "curly-hair 1280 32"; do
  n=${cfg%% *}
  for lib in product leaf4 refill8 refill24 susp8 susp24 susp32 product; do
    L=tools/_ab/libyhair_$lib.so; [ $lib = product ] && L=yocto-hair_amd/libyhair.so
    YHAIR_LIB=$L timeout -k 10 400 python3 tools/shape_check.py $cfg 3 2>&1 | grep Msamples | tail -1 | sed "s/^/$lib: /" | tee -a $out/ab_$n.txt
  done
  for P in 128 256; do
    YHAIR_ST_SLOTS=$P timeout -k 10 400 python3 tools/shape_check.py $cfg 3 2>&1 | grep Msamples | tail -1 | sed "s/^/slots$P: /" | tee -a $out/ab_$n.txt
  done
done
