#!/bin/bash
# Round 5: k_pool (launch shape 9) — bitwise check on the small scenes first, under a short timeout; then the dense configs against shapes 1 and 3.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5pool}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
WF_SHAPE=9 timeout -k 10 120 python3 tools/shape_check.py check 2>&1 | grep -v amdgpu.ids | tee $out/check9.txt || { echo "check FAILED rc=$?"; exit 1; }
grep -q "False" $out/check9.txt && { echo "NOT bit-identical"; exit 1; }
for cfg in "hair-curls 1280 32 1,9,3" "straight-hair 720 64 1,9,3" "curly-hair 1280 32 1,9,3" "sphere-hairblock 720 64 0,9"; do
  set -- $cfg
  timeout -k 10 300 python3 tools/shape_check.py $1 $2 $3 $4 2>&1 | grep -v amdgpu.ids | tee -a $out/perf9.txt || { echo "perf FAILED rc=$? on $1"; exit 1; }
done
