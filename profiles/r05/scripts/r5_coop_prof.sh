#!/bin/bash
# Round 5: per-branch counters of k_stream's trace stage (YHAIR_ST_PROF), cooperative line leaves against the lane's own.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-coopprof}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for cfg in "curly-hair 1280 16 3" "straight-hair 720 48 3" "hair-curls 1280 16 3"; do
  set -- $cfg
  for v in coop own; do
    YHAIR_ST_PROF=1 YHAIR_LIB=tools/_ab/libyhair_$v.so timeout -k 10 300 python3 tools/shape_check.py $cfg > $out/prof_${v}_$1.out 2> $out/prof_${v}_$1.txt || exit 1
    echo "== $v $cfg"; tail -30 $out/prof_${v}_$1.txt
  done
done
