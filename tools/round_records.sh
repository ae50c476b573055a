#!/bin/bash
# The record set of a round (one gpurun call): GPU tests, the bench line (no flags, the driver's flags, under rocprofv3, two ranks), every config with its CPU leg.
# usage: tools/round_records.sh TAG [a|b|all]    (a: tests + the C1 lines, b: two ranks + C2 / C3 / C4 — the whole set takes about 19 minutes, a gpurun call at most 20;
# the kernel-trial record on disk: a directory of this run's own, so that no leg sees another process's trials except
# where that is the point — ADVICE r04)
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-records}; mkdir -p $out; export TMPDIR=/tmp
export YHAIR_CACHE_DIR=/tmp/yhair_records_cache_$$; rm -rf $YHAIR_CACHE_DIR
part=${2:-all}
if [ $part != b ]; then
(time timeout -k 10 900 python -m pytest tests -m gpu -q --durations=8) > $out/gputests.log 2>&1; tail -4 $out/gputests.log
grep -q " passed" $out/gputests.log || exit 1
grep -q " failed" $out/gputests.log && exit 1
timeout -k 10 900 python3 bench.py > $out/bench_C1_no_flags.json 2> $out/bench_C1_no_flags.err || { tail -5 $out/bench_C1_no_flags.err; exit 1; }
rm -rf $YHAIR_CACHE_DIR   # the driver's run starts on a fresh box: no record
(time timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_C1_driver_flags.json 2> $out/bench_C1_driver_flags.err) 2> $out/bench_C1_driver_flags.time || exit 1
(cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-end-to-end > $GRAFT_REPO_ROOT/$out/bench_C1_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.log) || { tail -5 $out/rocprof.log; exit 1; }
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/C1_default_cmd_kernel_stats.csv \; ; find $out/prof -name "*kernel_trace.csv" -exec cp {} $out/C1_default_cmd_kernel_trace.csv \; ; rm -rf $out/prof
fi
if [ $part != a ]; then
timeout -k 10 900 python3 bench.py --gpus 2 > $out/bench_gpus2_bare.json 2> $out/bench_gpus2_bare.err || { tail -5 $out/bench_gpus2_bare.err; exit 1; }
for bm in 0.1 0.25 0.6; do timeout -k 10 900 python3 bench.py --config C2 --beta-m $bm --steps 8 > $out/bench_C2_betam$bm.json 2> $out/bench_C2_betam$bm.err || exit 1; done
for c in C3 C4; do timeout -k 10 900 python3 bench.py --config $c --steps 8 > $out/bench_$c.json 2> $out/bench_$c.err || exit 1; done
fi
rm -rf $YHAIR_CACHE_DIR
python3 - $out <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'unreadable', e); continue
    r=d['roofline']; p=d.get('parity') or {}
    print(os.path.basename(f), d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], r['kernel'][:28], 'launches', r.get('launches_in_timed_steps'), 'parity', p.get('spp'), p.get('ratio_to_floor', p.get('bitwise_equal')), (d.get('cpu_baseline') or {}).get('value'))
    oc=d['config'].get('other_configs')
    if oc:
        for k,v in oc['runs'].items(): print('   ', k, v if 'error' in v else (v['value'], v['ms_per_step'], v['kernel'][:20], v['launches_in_timed_steps'], v['roofline']['frac'], v['roofline']['traffic'], (v.get('parity') or {}).get('spp'), (v.get('parity') or {}).get('ratio_to_floor'), (v.get('cpu_baseline') or {}).get('value')))
    if 'path_following_light_hair' in p: print('   follow', {k:v for k,v in p['path_following_light_hair'].items() if k.startswith('ratio')})
    ps=d['config'].get('projected_strong_scaling')
    if ps: print('   projected', [(x.get('n_gpus'), x.get('value_if_every_gpu_takes_this_long'), (x.get('kernel') or '')[:24]) for x in ps['runs']])
    e=d['config'].get('end_to_end')
    if e: print('   end to end', {k:(e[k].get('total_wall_s'), e[k].get('sample_loop_s')) for k in ('cold_trial_record','warm_trial_record') if k in e}, e.get('largest_non_render_phase'), (e.get('reference_cli') or {}).get('projected_total_wall_s_at_full_spp'), e.get('error'))
PY
