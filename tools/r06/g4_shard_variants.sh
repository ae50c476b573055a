#!/bin/bash
# round 6, GPU call 4: the latency-side variants ON SHARDS — what one GPU of 4 / 8 renders — interleaved on one box:
#   lab        the product's device code (built from the lab worktree: the control), the whole resident grid
#   grid50/25  the same with half / a quarter of the resident workgroups (YHAIR_LAB_GRID_PCT: 2 and 1 waves per SIMD for the 256-thread forms)
#   prio       s_setprio 3 for the items taken by position (the head of the cost-sorted list), 0 for those from the cursor
#   touch      a lane that pushes a child of an 8- / 16-wide node touches the child's first line (consumed an iteration later)
# code: profiles/r06/shard_experiments.patch; harness: tools/shard_ab.py (min / median of 3 launches after 2 settling ones; md5 of the image)
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g4; mkdir -p $out; export TMPDIR=/tmp
L=$out/shard_variants.txt
run() {  # tag lib pct scene res spp full shapes
  TAG=$1 YHAIR_LIB=tools/_ab/libyhair_$2.so YHAIR_LAB_GRID_PCT=$3 timeout -k 10 240 python3 tools/shard_ab.py $4 $5 $6 $7 4,8 $8 2>&1 | grep -v amdgpu.ids | tee -a $L || echo "$1 $4 FAILED rc=$?" | tee -a $L
}
for r in 1 2; do
  echo "--- round $r" | tee -a $L
  for v in "lab lab 100" "grid50 lab 50" "grid25 lab 25" "prio prio 100" "touch touch 100"; do
    set -- $v
    run $1 $2 $3 sphere-hairblock 720 77 1536 8,6,4
    run $1 $2 $3 hair-curls 1280 256 4096 8,6,7
    [ $1 = touch ] || run $1 $2 $3 curly-hair 1280 256 4096 1,0
  done
done
# ---- the path slot regrouped by writer (item 3b): bitwise check, then k_stream on C3 / C2 against the lab build (old layout), interleaved
S=$out/slot_ab.txt
WF_SHAPE=3 timeout -k 10 120 python3 tools/shape_check.py check 2>&1 | grep -v amdgpu.ids | tee -a $S
for r in 1 2 3; do
  for cfg in "curly-hair 1280 64" "straight-hair 720 64"; do
    for v in lab product; do
      lib=tools/_ab/libyhair_$v.so; [ $v = product ] && lib=yocto-hair_amd/libyhair.so
      printf "%s r%s: " $v $r | tee -a $S
      YHAIR_NO_DISK_CACHE=1 YHAIR_LIB=$lib timeout -k 10 200 python3 tools/shape_check.py $cfg 3 2>&1 | grep Msamples | tail -1 | tee -a $S
    done
  done
done
