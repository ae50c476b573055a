#!/bin/bash
# Two builds of the library under the driver's bench command, interleaved. usage: r5_lib_bench_ab.sh TAG "name1 name2" [rounds]
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-libab}; mkdir -p $out; export TMPDIR=/tmp
for r in $(seq 1 ${3:-3}); do
  for v in $2; do
    YHAIR_LIB=tools/_ab/libyhair_$v.so timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-project-scaling --no-end-to-end --no-trial-cache > $out/$v$r.json 2> $out/$v$r.err || exit 1
    python3 - $out/$v$r.json $v $r <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'))
PY
  done
done
