#!/bin/bash
# round 6, GPU call 2: (a) GENERAL k_stream after the out-of-line lights pdf: bitwise check on the small scenes, then A/B against the round's first commit (tools/_ab/libyhair_r6base.so)
# on the scenes that run GENERAL kernels; (b) the chain profile (instrumented kernels) of C4 at one GPU and of the shards 0 of 8
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g2; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
G=$out/general_ab.txt
for s in 3 1; do
  echo "--- product, shape $s against shape 0 on the seven small scenes" | tee -a $G
  WF_SHAPE=$s timeout -k 10 120 python3 tools/shape_check.py check 2>&1 | grep -v amdgpu.ids | tee -a $G || { echo "check FAILED" | tee -a $G; exit 1; }
done
for r in 1 2; do
  for sc in volumes lobes textured; do
    for v in r6base product; do
      lib=tools/_ab/libyhair_$v.so; [ $v = product ] && lib=yocto-hair_amd/libyhair.so
      printf "%s r%s: " $v $r | tee -a $G
      YHAIR_LIB=$lib timeout -k 10 200 python3 tools/shape_check.py $sc 720 64 3,1 2>&1 | grep Msamples | tail -2 | tr '\n' ' ' | tee -a $G; echo | tee -a $G
    done
  done
done
C=$out/chain.txt
timeout -k 10 300 python3 tools/chain_profile.py hair-curls 1280 1 1 64 2>&1 | grep -v amdgpu.ids | tee -a $C
timeout -k 10 300 python3 tools/chain_profile.py hair-curls 1280 1,4,6 8 64 2>&1 | grep -v amdgpu.ids | tee -a $C
timeout -k 10 300 python3 tools/chain_profile.py curly-hair 1280 1,0 8 64 2>&1 | grep -v amdgpu.ids | tee -a $C
timeout -k 10 300 python3 tools/chain_profile.py sphere-hairblock 720 0,4,6 8 77 2>&1 | grep -v amdgpu.ids | tee -a $C
timeout -k 10 300 python3 tools/chain_profile.py sphere-hairblock 720 0,4,6 4 77 2>&1 | grep -v amdgpu.ids | tee -a $C
