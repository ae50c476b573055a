#!/bin/bash
# Round 5, VERDICT r04 item 2a: the touch-next-entry variants of k_stream (profiles/r04/touch_experiment/touch.patch, built as tools/_ab/libyhair_touch{0,1,2}.so:
# 0 = the product's code, 1 = a leaf step touches the wide node below it on the stack, 2 = a leaf below it too). Small scenes first, every run its own
# process under a timeout, output streamed.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5touch}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for t in 1 2; do
  echo "--- touch$t: bitwise check on the small scenes" | tee -a $out/touch.txt
  YHAIR_LIB=tools/_ab/libyhair_touch$t.so WF_SHAPE=3 timeout -k 10 60 python3 tools/shape_check.py check 2>&1 | grep -v amdgpu.ids | tee -a $out/touch.txt || { echo "touch$t check FAILED rc=$?" | tee -a $out/touch.txt; exit 1; }
done
grep -q "False" $out/touch.txt && { echo "NOT bit-identical" | tee -a $out/touch.txt; exit 1; }
for r in 1 2 3; do
  for cfg in "curly-hair 1280 32" "straight-hair 720 64"; do
    set -- $cfg
    for t in 0 1 2; do
      printf "touch%s r%s: " $t $r | tee -a $out/touch.txt
      YHAIR_LIB=tools/_ab/libyhair_touch$t.so timeout -k 10 90 python3 tools/shape_check.py $1 $2 $3 3 2>&1 | grep Msamples | tail -1 | tee -a $out/touch.txt || { echo "touch$t FAILED rc=$?" | tee -a $out/touch.txt; exit 1; }
    done
  done
done
