#!/bin/bash
# Round 5: k_pool's compile-time knobs, interleaved on one box (tools/build_variants.sh built the libraries).
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5poolsweep}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for cfg in "hair-curls 1280 32" "straight-hair 720 64" "curly-hair 1280 32"; do
  set -- $cfg
  for v in qp_base qp_w5 qp_r1 qp_r2 qp_f4 qp_f12 qp_fin48 qp_r2w5; do
    for P in 64 128; do
      printf "%s P=%s: " $v $P | tee -a $out/sweep.txt
      YHAIR_LIB=tools/_ab/libyhair_$v.so YHAIR_QP_SLOTS=$P timeout -k 10 200 python3 tools/shape_check.py $1 $2 $3 9 2>&1 | grep Msamples | tail -1 | tee -a $out/sweep.txt || exit 1
    done
  done
done
