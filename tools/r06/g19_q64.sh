#!/bin/bash
# round 6, GPU call 19: 64-byte quantised 4-wide nodes in k_stream's step (LAB: tools/_ab/libyhair_q64.so = profiles/r06/q64_nodes.patch) against the product, C3 and C2 at one GPU,
# shape 3 forced in both. The lab's boxes are conservative (a superset of the reference's children): NOT the product's bits — the images are compared numerically.
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g19; mkdir -p $out; export TMPDIR=/tmp
L=$out/q64.txt
run() { timeout -k 10 300 python3 tools/lab_render.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $L; }
# small first: the lab kernel on C1's scene at 180^2 (a fault here ends the call before the big scenes)
YHAIR_LIB=tools/_ab/libyhair_q64.so run sphere-hairblock 180 16 2 3 $out/c1_q64.npy &&
run sphere-hairblock 180 16 2 3 $out/c1_prod.npy &&
python3 tools/lab_compare.py $out/c1_q64.npy $out/c1_prod.npy | tee -a $L &&
for i in 1 2; do
  run curly-hair 1280 64 3 3 $out/c3_prod.npy &&
  YHAIR_LIB=tools/_ab/libyhair_q64.so run curly-hair 1280 64 3 3 $out/c3_q64.npy || exit 1
done
python3 tools/lab_compare.py $out/c3_q64.npy $out/c3_prod.npy | tee -a $L
for i in 1 2; do
  BETA_M=0.25 run straight-hair 720 192 3 3 $out/c2_prod.npy &&
  BETA_M=0.25 YHAIR_LIB=tools/_ab/libyhair_q64.so run straight-hair 720 192 3 3 $out/c2_q64.npy || exit 1
done
python3 tools/lab_compare.py $out/c2_q64.npy $out/c2_prod.npy | tee -a $L
rm -f $out/*.npy
