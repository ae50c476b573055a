#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4t; mkdir -p $out; export TMPDIR=/tmp
for cfg in "straight-hair 720 64 3" "curly-hair 1280 32 3"; do
  n=${cfg%% *}; set -- $cfg
  for lib in product stw3 policy1 lstack8 product; do
    L=tools/_ab/libyhair_$lib.so; [ $lib = product ] && L=yocto-hair_amd/libyhair.so
    YHAIR_LIB=$L timeout -k 10 400 python3 tools/shape_check.py $1 $2 $3 $4 2>&1 | grep Msamples | tail -1 | sed "s/^/$lib: /" | tee -a $out/ab_$n.txt
  done
  YHAIR_ST_WAVES=12 YHAIR_LIB=tools/_ab/libyhair_stw3.so timeout -k 10 400 python3 tools/shape_check.py $1 $2 $3 $4 2>&1 | grep Msamples | tail -1 | sed "s/^/stw3 (12 waves per CU): /" | tee -a $out/ab_$n.txt
done
bash tools/r4_final.sh r4final2
