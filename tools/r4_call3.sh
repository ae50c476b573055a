#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4c; mkdir -p $out; export TMPDIR=/tmp
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5) > $out/pytest.log 2>&1; tail -12 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
python3 tools/shape_check.py check > $out/check.txt 2>&1; cat $out/check.txt | tail -8
for cfg in "straight-hair 720 64" "curly-hair 1280 32" "hair-curls 1280 32"; do
  n=${cfg%% *}
  for lib in blob0 product blob0 product; do
    L=tools/_ab/libyhair_$lib.so; [ $lib = product ] && L=yocto-hair_amd/libyhair.so
    YHAIR_LIB=$L timeout -k 10 400 python3 tools/shape_check.py $cfg 3 2>&1 | grep Msamples | tail -1 | sed "s/^/$lib: /" | tee -a $out/ab_$n.txt
  done
done
YHAIR_ST_PROF=1 timeout -k 10 400 python3 tools/shape_check.py curly-hair 1280 32 3 > $out/prof_curly-hair.txt 2>&1; grep -A18 "k_stream" $out/prof_curly-hair.txt | tail -19
python3 tools/trace_only_bench.py > $out/trace_only.txt 2>&1; tail -12 $out/trace_only.txt
