#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4g; mkdir -p $out; export TMPDIR=/tmp
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5) > $out/pytest.log 2>&1; tail -12 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
for cfg in "sphere-hairblock 720 64 0,5" "hair-curls 1280 32 1" "straight-hair 720 64 1"; do
  n=${cfg%% *}; set -- $cfg
  for lib in qblob0 product qblob0 product; do
    L=tools/_ab/libyhair_$lib.so; [ $lib = product ] && L=yocto-hair_amd/libyhair.so
    YHAIR_LIB=$L timeout -k 10 400 python3 tools/shape_check.py $1 $2 $3 $4 2>&1 | grep Msamples | tail -2 | sed "s/^/$lib: /" | tee -a $out/ab_$n.txt
  done
done
( time YHAIR_TIMING=1 timeout -k 10 900 python3 bench.py > $out/bench_1.json 2> $out/bench_1.err ) 2>&1 | grep real
python3 - $out/bench_1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['launches_in_timed_steps'])
for k,v in d['config']['other_configs']['runs'].items(): print(k, v if 'error' in v else (v['value'], v['ms_per_step'], v['kernel'], v['launches_in_timed_steps'], v['roofline']['frac']))
print(d['parity'].get('path_following_light_hair'))
PY
