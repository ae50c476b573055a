#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4i; mkdir -p $out; export TMPDIR=/tmp
(time timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5) > $out/pytest.log 2>&1; tail -5 $out/pytest.log
grep -q " passed" $out/pytest.log || exit 1
bash tools/profile_configs.sh r4prof "C1:5:k_trace_sbs<false>" "C2b:3:k_stream" 2>&1 | tail -12
find gpurun_out/r4prof -name "*.db" -delete; find gpurun_out/r4prof -name "*agent_info*" -delete; du -sh gpurun_out/r4prof
