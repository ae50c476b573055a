#!/bin/bash
# round 6, GPU call 7: the records of the final tree — the driver's command bare and under rocprofv3 --kernel-trace --stats, every config's own line,
# then the counter passes of the four configs (tools/profile_round.sh)
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g7; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_C1_driver_flags.json 2> $out/bench_C1.err || echo "bench FAILED rc=$?"
tail -c 1800 $out/bench_C1_driver_flags.json; echo
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-project-scaling --no-end-to-end > $GRAFT_REPO_ROOT/$out/bench_C1_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/stats.log) || { tail -5 $out/stats.log; echo "rocprof FAILED"; }
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/C1_default_cmd_kernel_stats.csv \;
find $out/stats -name "*kernel_trace.csv" -exec cp {} $out/C1_default_cmd_kernel_trace.csv \;
rm -rf $out/stats; head -5 $out/C1_default_cmd_kernel_stats.csv
echo "--- per-config lines"
timeout -k 10 300 python3 bench.py --config C2 --beta-m 0.25 --steps 8 --no-other-configs --no-end-to-end > $out/bench_C2_betam0.25.json 2>/dev/null || echo "C2 line failed"
timeout -k 10 400 python3 bench.py --config C3 --steps 8 --no-other-configs --no-end-to-end > $out/bench_C3.json 2>/dev/null || echo "C3 line failed"
timeout -k 10 400 python3 bench.py --config C4 --steps 8 --no-other-configs --no-end-to-end > $out/bench_C4.json 2>/dev/null || echo "C4 line failed"
for f in C2_betam0.25 C3 C4; do python3 -c "
import json,sys; d=json.loads(open('$out/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['roofline']['frac'], d['roofline']['kernel'], [ (r['n_gpus'], r['seconds'], r['value_if_every_gpu_takes_this_long'], r['kernel']) for r in d['config'].get('projected_strong_scaling',{}).get('runs',[]) if 'seconds' in r])"; done
echo "--- counter passes"
bash tools/profile_round.sh g7pmc 2>&1 | tail -12
