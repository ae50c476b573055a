#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4j; mkdir -p $out; export TMPDIR=/tmp
(for args in "256 1000 300 44 4" "256 1000 350 44 4" "256 1000 400 44 4" "256 1000 450 44 4" "64 1000 400 44 4"; do timeout -k 5 120 tools/_ab/gather128 $args || break; done) > $out/gather128_mix.txt 2>&1
grep -E "^#|mode [01789]" $out/gather128_mix.txt
bash tools/profile_configs.sh r4prof "C3:3:k_stream" "C4:1:k_trace<false, false, 256, 5" 2>&1 | tail -12
find gpurun_out/r4prof -name "*.db" -delete; find gpurun_out/r4prof -name "*agent_info*" -delete; du -sh gpurun_out/r4prof
