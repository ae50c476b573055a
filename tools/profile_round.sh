#!/bin/bash
# The counter passes of the four BASELINE configs on the final device code (tools/profile_configs.sh), one gpurun call.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/profile_configs.sh ${1:-pmc} "C1:5:k_trace_sbs<false>" "C2b:3:k_stream<" "C3:3:k_stream<" "C4:1:k_trace<false, false, 256, 5"
