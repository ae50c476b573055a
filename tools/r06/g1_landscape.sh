#!/bin/bash
# round 6, GPU call 1: (a) the landscape of launch shapes on shards 0 of 4 / 0 of 8 of C3, C4, C1 with the product library; (b) the bench line with the new scalar keys
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g1; mkdir -p $out; export TMPDIR=/tmp
L=$out/landscape.txt
timeout -k 10 300 python3 tools/shard_ab.py curly-hair 1280 256 4096 4,8 0,1,3,3:slots=64,3:slots=192,4,7,6,8 2>&1 | grep -v amdgpu.ids | tee -a $L || echo "C3 landscape FAILED rc=$?" | tee -a $L
timeout -k 10 300 python3 tools/shard_ab.py hair-curls 1280 256 4096 4,8 0,1,3,3:slots=64,4,7,6,8 2>&1 | grep -v amdgpu.ids | tee -a $L || echo "C4 landscape FAILED rc=$?" | tee -a $L
timeout -k 10 200 python3 tools/shard_ab.py sphere-hairblock 720 77 1536 4,8 0,1,5,4,7,6,8 5 2>&1 | grep -v amdgpu.ids | tee -a $L || echo "C1 landscape FAILED rc=$?" | tee -a $L
BETA_M=0.25 timeout -k 10 200 python3 tools/shard_ab.py straight-hair 720 192 1536 4,8 0,1,3,3:slots=64,4,7,6,8 2>&1 | grep -v amdgpu.ids | tee -a $L || echo "C2 landscape FAILED rc=$?" | tee -a $L
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_C1.json 2> $out/bench_C1.err || echo "bench FAILED rc=$?"
tail -c 1500 $out/bench_C1.json
