#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r4e; mkdir -p $out
(for args in "2 1000 0 44 4" "2 1000 300 44 4" "2 1000 450 44 4" "16 1000 0 44 4" "64 1000 0 44 4" "64 1000 450 44 4" "256 1000 450 64 4"; do timeout -k 5 120 tools/_ab/gather128 $args || break; done) > $out/gather128_small.txt 2>&1
cat $out/gather128_small.txt
