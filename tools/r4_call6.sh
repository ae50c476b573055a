#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4f; mkdir -p $out; export TMPDIR=/tmp
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5) > $out/pytest.log 2>&1; tail -12 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
for cfg in "straight-hair 720 64" "curly-hair 1280 32" "hair-curls 1280 32"; do
  n=${cfg%% *}
  YHAIR_ST_PROF=1 timeout -k 10 400 python3 tools/shape_check.py $cfg 3 > $out/prof_$n.txt 2>&1 || { tail -5 $out/prof_$n.txt; exit 1; }
done
for i in 1 2; do
  ( time YHAIR_TIMING=1 timeout -k 10 900 python3 bench.py --no-cpu-baseline > $out/bench_$i.json 2> $out/bench_$i.err ) 2>&1 | grep real
  python3 - $out/bench_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['launches_in_timed_steps'], d['roofline']['kernel_trials'])
for k,v in d['config']['other_configs']['runs'].items(): print(k, v if 'error' in v else (v['value'], v['ms_per_step'], v['kernel'], v['launches_in_timed_steps'], v['roofline']['frac']))
print([ (r['n_gpus'], r['value_if_every_gpu_takes_this_long'], r['kernel']) for r in d['config']['projected_strong_scaling']['runs']])
PY
  grep -c "read from" $out/bench_$i.err
done
cat ~/.cache/yhair/trials_v1.txt | cut -c1-200 > $out/trials_file.txt; wc -l $out/trials_file.txt
