#!/bin/bash
# k_stream: bit check, then pool-size sweep with the per-stage profile
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-s5}; mkdir -p $out
WF_SHAPE=3 timeout -k 10 300 python tools/wf_check.py check > $out/check.txt 2>&1; echo "check rc $?" >> $out/check.txt
grep -v amdgpu.ids $out/check.txt
grep -q "check rc 0" $out/check.txt || exit 1
grep -q "False" $out/check.txt && exit 1
for cfg in "straight-hair 720 64" "curly-hair 1280 32" "hair-curls 1280 32" "sphere-hairblock 720 64"; do
  for P in ${SLOTS:-128 256 384}; do
    YHAIR_ST_SLOTS=$P timeout -k 10 300 python tools/wf_check.py $cfg 3 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/P=$P /" | tee -a $out/speed.txt || exit 1
    YHAIR_ST_SLOTS=$P YHAIR_ST_PROF=1 timeout -k 10 300 python tools/wf_check.py $cfg 3 2>&1 | grep -v amdgpu.ids | tail -9 | head -8 | tee -a $out/prof.txt || exit 1
  done
done
