#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4n; mkdir -p $out
timeout -k 10 300 python3 tools/chain_profile.py sphere-hairblock 720 0,5 > $out/chain_c1.txt 2>&1; cat $out/chain_c1.txt | grep -v amdgpu
timeout -k 10 300 python3 tools/chain_profile.py hair-curls 1280 1 > $out/chain_c4.txt 2>&1; cat $out/chain_c4.txt | grep -v amdgpu
