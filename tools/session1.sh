#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
( time timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/s1/pytest.log 2>&1
tail -30 gpurun_out/s1/pytest.log
timeout 900 bash tools/bsdf_variants.sh gpurun_out/s1/bsdf "fast:" "exact:-DYH_HAIR_FAST=0" "asin:-DYH_FAST_ASIN=0" "div:-DYH_FAST_DIV=0" "asindiv:-DYH_FAST_ASIN=0 -DYH_FAST_DIV=0" "log:-DYH_FAST_LOG=0" > gpurun_out/s1/bsdf.log 2>&1
cat gpurun_out/s1/bsdf/summary.txt
timeout 300 python tools/tile_costs.py sphere-hairblock 64 > gpurun_out/s1/tile_C1.txt 2>&1
timeout 300 python tools/tile_costs.py straight-hair 64 > gpurun_out/s1/tile_C2.txt 2>&1
tail -12 gpurun_out/s1/tile_C1.txt gpurun_out/s1/tile_C2.txt
