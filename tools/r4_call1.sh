#!/bin/bash
# Round-4 GPU session 1 (one gpurun call): memory-side microbenchmark, available counters, k_stream branch profile, baselines.
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4a; mkdir -p $out; export TMPDIR=/tmp
(for args in "256 1000 600 44 4" "256 1000 0 44 4" "256 1000 300 44 4" "256 1000 600 64 4" "32 1000 600 44 4" "256 1000 600 44 2" "1024 1000 600 44 4"; do timeout -k 5 120 tools/_ab/gather128 $args || break; done) > $out/gather128.txt 2>&1
tail -8 $out/gather128.txt
(cd /tmp && timeout -k 5 120 rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$out/avail.txt 2>&1)
echo "avail lines: $(wc -l < $out/avail.txt)"
for cfg in "straight-hair 720 64" "curly-hair 1280 32" "hair-curls 1280 32"; do
  n=${cfg%% *}
  YHAIR_ST_PROF=1 timeout -k 10 400 python3 tools/shape_check.py $cfg 3 > $out/prof_$n.txt 2>&1 || { tail -5 $out/prof_$n.txt; exit 1; }
  grep -A18 "k_stream" $out/prof_$n.txt | tail -19
done
for cfg in "straight-hair 720 64" "curly-hair 1280 32" "hair-curls 1280 32"; do
  n=${cfg%% *}
  timeout -k 10 400 python3 tools/shape_check.py $cfg 1,3 > $out/base_$n.txt 2>&1 || { tail -5 $out/base_$n.txt; exit 1; }
  YHAIR_LIB=tools/_ab/libyhair_take16.so timeout -k 10 400 python3 tools/shape_check.py $cfg 3 > $out/take16_$n.txt 2>&1 || { tail -5 $out/take16_$n.txt; exit 1; }
  grep Msamples $out/base_$n.txt $out/take16_$n.txt
done
timeout -k 10 300 python3 tools/shape_check.py sphere-hairblock 720 64 0,5 > $out/base_c1.txt 2>&1; grep Msamples $out/base_c1.txt
