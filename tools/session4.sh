#!/bin/bash
# k_stream: throughput against k_trace on the BASELINE configs, then its per-stage profile (YHAIR_ST_PROF)
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-s4}; mkdir -p $out
for cfg in "straight-hair 720 64" "curly-hair 1280 32" "hair-curls 1280 32" "sphere-hairblock 720 64"; do
  timeout -k 10 300 python tools/wf_check.py $cfg ${SHAPES:-1,3} 2>&1 | grep -v amdgpu.ids | tee -a $out/speed.txt || exit 1
  YHAIR_ST_PROF=1 timeout -k 10 300 python tools/wf_check.py $cfg 3 2>&1 | grep -v amdgpu.ids | tail -12 | tee -a $out/prof.txt || exit 1
done
