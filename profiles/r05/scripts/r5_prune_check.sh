#!/bin/bash
# Round 5: the pruned tree — the whole GPU suite, then pruned against HEAD's library, interleaved, on the four configs.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5prune}; mkdir -p $out; export TMPDIR=/tmp
(time timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=8) > $out/gputests.log 2>&1; tail -6 $out/gputests.log
grep -q " passed" $out/gputests.log || exit 1
grep -q " failed" $out/gputests.log && exit 1
export YHAIR_NO_DISK_CACHE=1
for r in 1 2; do
  for cfg in "sphere-hairblock 720 64 5" "hair-curls 1280 32 1" "straight-hair 720 64 3" "curly-hair 1280 32 3"; do
    set -- $cfg
    for v in head pruned; do
      lib=tools/_ab/libyhair_head.so; [ $v = pruned ] && lib=yocto-hair_amd/libyhair.so
      printf "%s r%s: " $v $r | tee -a $out/prune_ab.txt
      YHAIR_LIB=$lib timeout -k 10 200 python3 tools/shape_check.py $1 $2 $3 $4 2>&1 | grep Msamples | tail -1 | tee -a $out/prune_ab.txt || exit 1
    done
  done
done
