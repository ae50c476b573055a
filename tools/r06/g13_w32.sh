#!/bin/bash
# round 6, GPU call 13: THIRTY-TWO lanes per path (32-wide nodes: five binary levels per step, leaf groups of up to eight leaves; LAB build, launch shape 9 forced):
# bitwise against the quad kernel on the seven small scenes, then against the sixteen-lane form with leaf groups (shape 8) on the shards and on C1 at 180^2
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g13; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
L=$out/w32.txt
YHAIR_LIB=tools/_ab/libyhair_w32.so WF_SHAPE=9 timeout -k 10 120 python3 tools/shape_check.py check 2>&1 | grep -v amdgpu.ids | tee -a $L || { echo "check FAILED" | tee -a $L; exit 1; }
grep -q "False" $L && { echo "NOT bit-identical: stopping" | tee -a $L; exit 1; }
for r in 1 2; do
  echo "--- round $r" | tee -a $L
  TAG=w32 YHAIR_LIB=tools/_ab/libyhair_w32.so timeout -k 10 240 python3 tools/shard_ab.py sphere-hairblock 720 77 1536 8,4,2 9,8 2>&1 | grep -v amdgpu.ids | tee -a $L
  TAG=w32 YHAIR_LIB=tools/_ab/libyhair_w32.so timeout -k 10 240 python3 tools/shard_ab.py hair-curls 1280 256 4096 8,4 9,8 2>&1 | grep -v amdgpu.ids | tee -a $L
  TAG=w32 YHAIR_LIB=tools/_ab/libyhair_w32.so timeout -k 10 240 python3 tools/shard_ab.py sphere-hairblock 180 77 1536 1 9,8 2>&1 | grep -v amdgpu.ids | tee -a $L
  TAG=w32 YHAIR_LIB=tools/_ab/libyhair_w32.so timeout -k 10 240 python3 tools/shard_ab.py straight-hair 720 192 1536 8 9,8,4 2>&1 | grep -v amdgpu.ids | tee -a $L
done
