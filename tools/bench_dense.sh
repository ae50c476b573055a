#!/bin/bash
# Throughput of the dense-hair configs C2..C4 (BASELINE.json configs[2..4]) through bench.py, no CPU leg.
# usage: tools/bench_dense.sh [outdir]   (run on the GPU box)
out=${1:-gpurun_out/dense}; mkdir -p "$out"
python bench.py --no-cpu-baseline --scene straight-hair --resolution 720 --spp-per-step 64 --steps 6 --warmup 2 > "$out/C2.json"
python bench.py --no-cpu-baseline --scene curly-hair --resolution 1280 --spp-per-step 32 --steps 6 --warmup 2 > "$out/C3.json"
python bench.py --no-cpu-baseline --scene hair-curls --resolution 1280 --spp-per-step 32 --steps 6 --warmup 2 > "$out/C4.json"
for c in C2 C3 C4; do python3 -c "
import json,sys; d=json.loads(open('$out/$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['unit'], 'ms/step', d['ms_per_step'])"; done
