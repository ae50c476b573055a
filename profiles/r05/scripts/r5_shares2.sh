#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5shares2}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
YHAIR_TIMING=1 YHAIR_ST_WAVELOG=1 timeout -k 10 400 python3 tools/launch_series.py straight-hair 720 192 10 3 2>&1 | grep "wave log\|dispatch round\|correlation\|wave-slot speeds\|launches:" | tee -a $out/series.txt
