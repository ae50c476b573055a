#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4b; mkdir -p $out; export TMPDIR=/tmp
(for args in "256 1000 600 44 4" "256 1000 600 44 6" "256 1000 600 44 8" "256 1000 300 44 8" "256 1000 0 44 8" "256 1000 450 44 4" "256 1000 450 44 8"; do timeout -k 5 120 tools/_ab/gather128_w8 $args || break; done) > $out/gather128_w8.txt 2>&1
grep -E "^#|mode [036]" $out/gather128_w8.txt
for cfg in "straight-hair 720 64" "curly-hair 1280 32" "hair-curls 1280 32"; do
  n=${cfg%% *}
  YHAIR_ST_PROF=1 timeout -k 10 400 python3 tools/shape_check.py $cfg 3 > $out/prof_$n.txt 2>&1 || { tail -5 $out/prof_$n.txt; exit 1; }
  grep -A18 "k_stream" $out/prof_$n.txt | tail -19
done
( time timeout -k 10 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['launches_in_timed_steps'], d['roofline']['algorithmic_bytes_per_sample'])
for k,v in d['config']['other_configs']['runs'].items(): print(k, v if 'error' in v else (v['value'], v['ms_per_step'], v['kernel'], v['launches_in_timed_steps'], v['roofline']['frac'], v['roofline']['traffic'], v['roofline']['traffic_source']))
print(d.get('parity')); print(d.get('cpu_baseline'))
PY
