#!/bin/bash
# round 6, GPU call 9: the final tree — whole GPU suite, smoke, the driver's command (bare, timed, and under rocprofv3), the counter passes again (the device sources changed after g7)
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g9; mkdir -p $out; export TMPDIR=/tmp
(time timeout -k 10 900 python3 -m pytest tests -q -m gpu --durations=6) > $out/gputests.log 2>&1; tail -12 $out/gputests.log
timeout -k 10 300 python3 __graft_entry__.py smoke 2>&1 | tail -2 | tee $out/smoke.txt
(time timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_C1_driver_flags.json 2> $out/bench_C1.err) 2> $out/bench_C1_driver_flags.time || echo "bench FAILED rc=$?"
cat $out/bench_C1_driver_flags.time | tail -3; tail -c 1700 $out/bench_C1_driver_flags.json; echo
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-project-scaling --no-end-to-end > $GRAFT_REPO_ROOT/$out/bench_C1_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/stats.log) || { tail -5 $out/stats.log; echo "rocprof FAILED"; }
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/C1_default_cmd_kernel_stats.csv \;
find $out/stats -name "*kernel_trace.csv" -exec cp {} $out/C1_default_cmd_kernel_trace.csv \;
rm -rf $out/stats; head -3 $out/C1_default_cmd_kernel_stats.csv
bash tools/profile_round.sh g9pmc 2>&1 | grep "done\|fail\|rror" | tail -8
