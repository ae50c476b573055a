#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout 300 python tools/wf_check.py check > gpurun_out/s2/check.txt 2>&1; echo "check rc $?" >> gpurun_out/s2/check.txt
cat gpurun_out/s2/check.txt
grep -q "check rc 0" gpurun_out/s2/check.txt || exit 1
for cfg in "straight-hair 720 64" "sphere-hairblock 720 64" "curly-hair 1280 32" "hair-curls 1280 32"; do
  timeout 300 python tools/wf_check.py $cfg 0,1,2 2>&1 | tee -a gpurun_out/s2/speed.txt
done
YHAIR_WF_SLOTS=2 timeout 300 python tools/wf_check.py straight-hair 720 64 2 2>&1 | tee -a gpurun_out/s2/speed.txt
YHAIR_WF_SLOTS=2 timeout 300 python tools/wf_check.py sphere-hairblock 720 64 2 2>&1 | tee -a gpurun_out/s2/speed.txt
