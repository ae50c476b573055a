#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4s; mkdir -p $out; export TMPDIR=/tmp
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5) > $out/pytest.log 2>&1; tail -4 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
rm -rf gpurun_out/r4prof2
bash tools/profile_configs.sh r4prof2 "C1:5:k_trace_sbs<false>" "C2b:3:k_stream" "C3:3:k_stream" "C4:1:k_trace<false, false, 256, 5" 2>&1 | grep "== \|k_trace\|k_stream" | head -20
find gpurun_out/r4prof2 -name "*.db" -delete; find gpurun_out/r4prof2 -name "*agent_info*" -delete; du -sh gpurun_out/r4prof2
