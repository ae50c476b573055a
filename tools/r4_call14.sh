#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4m; mkdir -p $out; export TMPDIR=/tmp
for K in 0 16 32 44 64 96 128 44 32; do
  if [ $K = 0 ]; then unset YHAIR_HY_OCT; else export YHAIR_HY_OCT=$K; fi
  timeout -k 10 400 python3 tools/shape_check.py sphere-hairblock 720 64 5 2>&1 | grep Msamples | tail -1 | sed "s/^/K=$K: /" | tee -a $out/sbs_k.txt
done
unset YHAIR_HY_OCT
timeout -k 10 900 python3 bench.py --no-other-configs --no-project-scaling > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['parity'].get('path_following_light_hair'))"
