#!/bin/bash
# Developer tool: builds k_trace variants (-D overrides) on the GPU box and benches each.
# usage: tools/kernel_sweep.sh "name1:-DYH_MIN_WAVES=2" "name2:-DYH_LDS_STACK=16 ..." -- <bench args>
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
variants=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p /tmp/yh_sweep
for v in "${variants[@]}"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -fPIC -std=c++17 -I$R/include -I$R/yocto-hair_amd/csrc $flags \
      -c $R/yocto-hair_amd/csrc/kernels.hip -o /tmp/yh_sweep/k_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/yh_sweep/libyhair_$name.so /tmp/yh_sweep/k_$name.o $R/yocto-hair_amd/csrc/bvh_gpu.o \
      $R/yocto-hair_amd/host/context.o $R/yocto-hair_amd/host/bvh_build.o $R/yocto-hair_amd/host/scene_io.o -lpthread -lz
  echo "== $name ($flags) env: ${YH_ENV}"
  YHAIR_LIB=/tmp/yh_sweep/libyhair_$name.so python3 $R/bench.py --no-cpu-baseline "$@" 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|Error.*\|error.*' | tr '\n' ' '
  echo
done
