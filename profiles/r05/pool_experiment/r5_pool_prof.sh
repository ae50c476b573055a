#!/bin/bash
# Round 5: where k_pool's wave time goes (YHAIR_ST_PROF=1), per slots-per-wave setting, on C4 and C2.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5poolprof}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for cfg in "hair-curls 1280 32" "straight-hair 720 64"; do
  set -- $cfg
  for P in 64 128 192 256; do
    echo "=== $1 slots $P" | tee -a $out/prof.txt
    YHAIR_QP_SLOTS=$P timeout -k 10 200 python3 tools/shape_check.py $1 $2 $3 9 2>&1 | grep -v amdgpu.ids | tee -a $out/prof.txt || exit 1
    YHAIR_ST_PROF=1 YHAIR_QP_SLOTS=$P timeout -k 10 200 python3 tools/shape_check.py $1 $2 $3 9 2>&1 | grep -v amdgpu.ids | grep -A8 "k_pool" | tail -9 | tee -a $out/prof.txt || exit 1
  done
done
