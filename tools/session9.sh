#!/bin/bash
# k_stream item groups per XCD: bit check, then G = 1 against G = 8 on the dense configs
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-s12}; mkdir -p $out
WF_SHAPE=3 timeout -k 10 300 python tools/wf_check.py check > $out/check.txt 2>&1; echo "check rc $?" >> $out/check.txt
grep -v amdgpu.ids $out/check.txt
grep -q "check rc 0" $out/check.txt || exit 1
grep -q "False" $out/check.txt && exit 1
for cfg in "curly-hair 1280 32" "straight-hair 720 64" "hair-curls 1280 32"; do
  for G in 1 8 1 8; do
    YHAIR_ST_GROUPS=$G timeout -k 10 300 python tools/wf_check.py $cfg 3 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/G=$G /" | tee -a $out/speed.txt || exit 1
  done
done
