#!/bin/bash
# Developer tool (GPU box): where a step of k_stream's trace stage spends its time — the instrumented build's five stamps per step (YHAIR_ST_PROF=1,
# csrc/dev_lane.h: stamp) on C3, C2 and k_stream forced on C4. usage: tools/stream_step_parts.sh TAG
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-stepparts}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for cfg in "curly-hair 1280 16 3" "straight-hair 720 48 3" "hair-curls 1280 16 3"; do
  set -- $cfg
  YHAIR_ST_PROF=${PROFMODE:-2} timeout -k 10 300 python3 tools/shape_check.py $cfg > $out/$1.out 2> $out/$1.txt || exit 1
  echo "== $cfg"; grep "k_stream\|trace\|step part" $out/$1.txt | tail -9
done
