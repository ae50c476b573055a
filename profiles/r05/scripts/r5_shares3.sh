#!/bin/bash
# Round 5: k_stream's shares by wave-slot speed with the clean kernel (no spill in the step loop): series of launches on C2, the old hand-out
# (YHAIR_ST_SLOTS=128 forces the round-4 geometry and dealing) against the shares; C3 and the small scenes as a sanity check.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5shares5}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
WF_SHAPE=3 timeout -k 10 120 python3 tools/shape_check.py check 2>&1 | grep -c "images equal True  rng equal True" | tee -a $out/series.txt
timeout -k 10 300 python3 tools/shape_check.py straight-hair 720 64 1,3 2>&1 | grep Msamples | tee -a $out/series.txt || exit 1
for r in 1 2; do
  echo "--- old hand-out (YHAIR_ST_SLOTS=128) r$r" | tee -a $out/series.txt
  YHAIR_ST_SLOTS=128 YHAIR_ST_WAVELOG=1 timeout -k 10 400 python3 tools/launch_series.py straight-hair 720 192 10 3 2>&1 | grep "launches:\|wave log" | tail -3 | tee -a $out/series.txt
  echo "--- shares r$r" | tee -a $out/series.txt
  YHAIR_TIMING=1 YHAIR_ST_WAVELOG=1 timeout -k 10 400 python3 tools/launch_series.py straight-hair 720 192 10 3 2>&1 | grep "launches:\|wave log\|shares:" | tail -5 | tee -a $out/series.txt
done
timeout -k 10 300 python3 tools/shape_check.py curly-hair 1280 32 3 2>&1 | grep Msamples | tee -a $out/series.txt
