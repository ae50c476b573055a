#!/bin/bash
# round 6, GPU call 22: the box test of the octet / sixteen-lane node steps with packed FP32 subtractions and multiplications (LAB: tools/_ab/libyhair_pkbox.so =
# profiles/r06/pkbox.patch) against the product on the shards where those kernels run at ~1 wave per SIMD. Same bits expected (md5 column).
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g22; mkdir -p $out; export TMPDIR=/tmp
L=$out/pkbox.txt
for i in 1 2; do
  for lib in product pkbox; do
    if [ $lib = product ]; then unset YHAIR_LIB; else export YHAIR_LIB=tools/_ab/libyhair_$lib.so; fi
    TAG=$lib timeout -k 10 200 python3 tools/shard_ab.py sphere-hairblock 720 256 1536 8 8,6,4 5 2>&1 | grep -v amdgpu.ids | tee -a $L || exit 1
    TAG=$lib timeout -k 10 200 python3 tools/shard_ab.py hair-curls 1280 256 4096 8 8,7 2>&1 | grep -v amdgpu.ids | tee -a $L || exit 1
    TAG=$lib timeout -k 10 200 python3 tools/shard_ab.py curly-hair 1280 256 4096 8 4,7 2>&1 | grep -v amdgpu.ids | tee -a $L || exit 1
  done
done
