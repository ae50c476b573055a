#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4k; mkdir -p $out; export TMPDIR=/tmp
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5) > $out/pytest.log 2>&1; tail -5 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
for cfg in "straight-hair 720 64 3" "curly-hair 1280 32 3" "hair-curls 1280 32 1" "sphere-hairblock 720 64 0,5"; do
  n=${cfg%% *}; set -- $cfg
  for v in noalign align noalign align; do
    if [ $v = noalign ]; then export YHAIR_NO_LEAF_ALIGN=1; else unset YHAIR_NO_LEAF_ALIGN; fi
    YHAIR_SCENES=/tmp/yhair_scenes timeout -k 10 400 python3 tools/shape_check.py $1 $2 $3 $4 2>&1 | grep Msamples | tail -2 | sed "s/^/$v: /" | tee -a $out/ab_$n.txt
  done
done
