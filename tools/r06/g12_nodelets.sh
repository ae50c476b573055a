#!/bin/bash
# round 6, GPU call 12: LDS nodelets re-examined WHERE LATENCY BINDS: the root and the sixteen children of the dominant hair shape's
# 16-wide tree (8.7 KB) staged in LDS by the sixteen-lane kernels (launch shapes 6, 8) — on shards 0 of 8 / 0 of 4 of C1 and C4 and on C1 at 180^2, interleaved
# with the control; 1 node (the root: 512 B) and 273 nodes (three levels: 140 KB does not fit -> skipped) as further points
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g12; mkdir -p $out; export TMPDIR=/tmp
L=$out/nodelets.txt
run() { TAG=$1 YHAIR_LIB=tools/_ab/libyhair_$2.so YHAIR_LAB_NL_NODES=$3 timeout -k 10 240 python3 tools/shard_ab.py $4 $5 $6 $7 $8 $9 2>&1 | grep -v amdgpu.ids | tee -a $L || echo "$1 $4 FAILED rc=$?" | tee -a $L; }
for r in 1 2 3; do
  echo "--- round $r" | tee -a $L
  for v in "control lab2 0" "nodelets17 nodelets 17" "nodelets1 nodelets 1"; do
    set -- $v
    run $1 $2 $3 sphere-hairblock 720 77 1536 4,8 8,6
    run $1 $2 $3 hair-curls 1280 256 4096 4,8 8,6
    run $1 $2 $3 sphere-hairblock 180 77 1536 1 8,6
  done
done
