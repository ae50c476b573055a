#!/bin/bash
# k_stream (YHAIR_SHAPE=3) against k_trace: bitwise check on small scenes, then throughput on the dense configs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
WF_SHAPE=3 timeout -k 10 300 python tools/wf_check.py check > gpurun_out/s3/check.txt 2>&1; echo "check rc $?" >> gpurun_out/s3/check.txt
cat gpurun_out/s3/check.txt
grep -q "check rc 0" gpurun_out/s3/check.txt || exit 1
for cfg in "straight-hair 720 64" "curly-hair 1280 32" "hair-curls 1280 32" "sphere-hairblock 720 64"; do
  timeout -k 10 300 python tools/wf_check.py $cfg 1,3 2>&1 | tee -a gpurun_out/s3/speed.txt || exit 1
done
