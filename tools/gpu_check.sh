#!/bin/bash
# Developer tool (GPU box): the GPU tests, then bench.py on the four BASELINE configs (kernel chosen by measurement). usage: tools/gpu_check.sh [outdir-name]
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-s6}; mkdir -p $out
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8) > $out/pytest.log 2>&1; tail -14 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
YHAIR_TIMING=1 timeout -k 10 600 python bench.py > $out/bench_C1.json 2> $out/bench_C1.err || { tail -20 $out/bench_C1.err; exit 1; }
cat $out/bench_C1.json; grep "kernel times\|launch shape" $out/bench_C1.err | tail -3
for c in C2 C3 C4; do
  YHAIR_TIMING=1 timeout -k 10 600 python bench.py --config $c --no-cpu-baseline --steps 8 > $out/bench_$c.json 2> $out/bench_$c.err || { tail -20 $out/bench_$c.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$out/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['unit'], 'ms/step', d['ms_per_step'], d['roofline']['kernel'], d['config']['workload'])"
  grep "kernel times" $out/bench_$c.err | tail -1
done
