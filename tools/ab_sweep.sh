#!/bin/bash
# Developer tool: A/B/A/B timing of k_trace build variants inside ONE gpurun call (box-to-box and
# run-to-run spread is several per cent, so variants are only comparable interleaved on one box).
# usage: tools/ab_sweep.sh ROUNDS "name:-Dflags" ... -- <bench args>
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=$1; shift
variants=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p /tmp/yh_sweep
for v in "${variants[@]}"; do
  name=${v%%:*}; flags=${v#*:}
  [ -f /tmp/yh_sweep/libyhair_$name.so ] && continue
  src=$R/yocto-hair_amd/csrc
  case $name in old*) src=$R/tools/_old_csrc;; esac   # a snapshot of an earlier csrc/ placed there by hand
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -fPIC -std=c++17 -I$R/include -I$src $flags \
      -c $src/kernels.hip -o /tmp/yh_sweep/k_$name.o &
done
wait
for v in "${variants[@]}"; do
  name=${v%%:*}
  [ -f /tmp/yh_sweep/libyhair_$name.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/yh_sweep/libyhair_$name.so /tmp/yh_sweep/k_$name.o $R/yocto-hair_amd/csrc/wide.o $R/yocto-hair_amd/csrc/exact.o $R/yocto-hair_amd/csrc/stream.o $R/yocto-hair_amd/csrc/bvh_gpu.o \
      $R/yocto-hair_amd/host/*.o -lpthread -lz -ldl
done
[ -n "${AB_PREBUILT:-}" ] && cp "$AB_PREBUILT" /tmp/yh_sweep/libyhair_prebuilt.so && variants+=("prebuilt:")   # e.g. AB_PREBUILT=tools/_ab/libyhair_<commit>.so
for r in $(seq 1 $rounds); do
  for v in "${variants[@]}"; do
    name=${v%%:*}
    printf "%s r%d: " "$name" "$r"
    YHAIR_LIB=/tmp/yh_sweep/libyhair_$name.so python3 $R/bench.py --no-cpu-baseline --no-project-scaling --no-other-configs "$@" 2>&1 | grep -o '"value": [0-9.]*' | tr '\n' ' '
    echo
  done
done
