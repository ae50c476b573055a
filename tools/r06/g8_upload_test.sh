#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g8; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "upload_on_the_device or wide_collapses or cli_matches or light_sampling or reference_scene_files" 2>&1 | tail -15 | tee $out/pytest_subset.txt
