#!/bin/bash
# round 6, GPU call 18: k_stream on an under-filled shard (C3 0 of 8: 204 800 paths for 262 144 lanes — bound by the time ONE sample takes through the stages): does leaving the
# trace stage earlier (suspend threshold 8 ... 48 busy lanes) or a smaller pool (64 / 128 slots per wave) shorten that? host-side parameters only (LAB: YHAIR_LAB_SUSPEND)
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g18; mkdir -p $out; export TMPDIR=/tmp
L=$out/stream_shard.txt
for susp in 16 8 32 48; do
  for w in "curly-hair 1280 256 4096 8" "curly-hair 1280 256 4096 4" "straight-hair 720 192 1536 8"; do
    set -- $w
    TAG=susp$susp YHAIR_LIB=tools/_ab/libyhair_susp.so YHAIR_LAB_SUSPEND=$susp timeout -k 10 240 python3 tools/shard_ab.py $1 $2 $3 $4 $5 3,3:slots=64 2>&1 | grep -v amdgpu.ids | tee -a $L
  done
done
