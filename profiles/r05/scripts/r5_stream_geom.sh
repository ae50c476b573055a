#!/bin/bash
# Round 5: k_stream on hair-curls (C4) with FEWER waves and MORE slots per wave. C4 has ~164 k expensive pixels for 262 k lanes: at 4 waves per SIMD a
# wave holds ~40 expensive paths (22 of 64 lanes busy); at 2 waves per SIMD it holds ~80.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5geom}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
scene=${2:-hair-curls}; res=${3:-1280}; spp=${4:-64}
for W in 16 12 8 4; do
  for P in 128 192 256 384 512; do
    printf "waves per CU %s slots %s: " $W $P | tee -a $out/geom_$scene.txt
    YHAIR_ST_WAVES=$W YHAIR_ST_SLOTS=$P timeout -k 10 200 python3 tools/shape_check.py $scene $res $spp 3 2>&1 | grep Msamples | tail -1 | tee -a $out/geom_$scene.txt || exit 1
  done
done
