#!/bin/bash
# Round profile set (GPU box): for each BASELINE config the rocprofv3 kernel statistics of the bench command and the
# PMC passes of its sample-loop kernel (tools/pmc_k_trace.sh). usage: tools/profile_configs.sh TAG "C1:0:k_trace<false, false, 512, 4>" ...
# (config : YHAIR_SHAPE : kernel-name filter)
cd "$(dirname "$0")/.." && ROOT=$PWD
export TMPDIR=/tmp
TAG=$1; shift
for spec in "$@"; do
  cfg=${spec%%:*}; rest=${spec#*:}; shape=${rest%%:*}; kern=${rest#*:}
  out=$ROOT/gpurun_out/$TAG/$cfg; mkdir -p $out
  steps=6; [ "$cfg" = C1 ] && steps=24
  spp=64; case $cfg in C3|C4) spp=32;; esac   # samples per launch, as in the round-1 profiles
  case $cfg in C1|C2|C3|C4) sel="--config $cfg";; C2b) sel="--config C2 --beta-m 0.25";; *) sel="--scene $cfg";; esac   # C2b = C2 at the middle of its beta_m sweep: what bench.py's config.other_configs runs   # (other names: a scene of tools/make_scenes.py at C1's image size)
  (cd /tmp && YHAIR_SHAPE=$shape timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $ROOT/bench.py $sel --spp-per-step $spp --no-cpu-baseline --no-project-scaling --no-other-configs --steps $steps > $out/bench_under_rocprof.json 2> $out/stats.log) || { tail -5 $out/stats.log; exit 1; }
  find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
  head -4 $out/kernel_stats.csv
  YHAIR_SHAPE=$shape PMC_KERNEL="$kern" PMC_ARGS="$sel --spp-per-step $spp --no-project-scaling --no-other-configs" bash tools/pmc_k_trace.sh $TAG/$cfg/pmc > $out/pmc.txt 2>&1 || exit 1
  cp $ROOT/gpurun_out/$TAG/$cfg/pmc/k_trace_pmc.json $out/pmc.json
  echo "== $cfg done"
done
