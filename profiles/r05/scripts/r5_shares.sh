#!/bin/bash
# Round 5: k_stream's shares by wave-slot speed (one-generation images) — same image as the quad kernel, the waves' end stamps, with / without.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5shares}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
timeout -k 10 300 python3 tools/shape_check.py straight-hair 720 64 1,3 2>&1 | grep Msamples | tee -a $out/shares.txt || exit 1
for r in 1 2 3; do
  for v in noshares shares; do
    printf "%s r%s: " $v $r | tee -a $out/shares.txt
    if [ $v = noshares ]; then export YHAIR_ST_NO_SHARES=1; else unset YHAIR_ST_NO_SHARES; fi
    YHAIR_ST_WAVELOG=1 timeout -k 10 200 python3 tools/shape_check.py straight-hair 720 192 3 2>&1 | grep "Msamples\|wave log\|dispatch round" | tail -6 | tee -a $out/shares.txt || exit 1
  done
done
unset YHAIR_ST_NO_SHARES
for P in 0 448; do
  echo "curly-hair slots $P (0 = default)" | tee -a $out/shares.txt
  if [ $P = 0 ]; then unset YHAIR_ST_SLOTS; else export YHAIR_ST_SLOTS=$P; fi
  YHAIR_ST_WAVELOG=1 timeout -k 10 300 python3 tools/shape_check.py curly-hair 1280 128 3 2>&1 | grep "Msamples\|wave log" | tail -3 | tee -a $out/shares.txt || exit 1
done
