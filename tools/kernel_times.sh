#!/bin/bash
# Developer tool (GPU box): the GPU tests, then the four configs with each kernel forced (tools/wf_check.py: bitwise comparison + timing). usage: tools/kernel_times.sh [outdir-name]
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-s8}; mkdir -p $out
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5) > $out/pytest.log 2>&1; tail -9 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
grep -q "failed" $out/pytest.log && exit 1
for cfg in "sphere-hairblock 720 64 0,1" "straight-hair 720 64 1,3" "curly-hair 1280 32 1,3" "hair-curls 1280 32 1,3"; do
  timeout -k 10 300 python tools/wf_check.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $out/speed.txt || exit 1
done
