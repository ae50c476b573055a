#!/bin/bash
# Developer tool (GPU box): path fidelity (oracle/divergence_report.py) and C1 throughput of k_trace built with
# different BSDF arithmetic switches (csrc/dev_hair.h: YH_HAIR_FAST, YH_FAST_DIV / _LOG / _TRIG / _ASIN).
# usage: tools/bsdf_variants.sh outdir "name:-Dflags" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$1; shift; mkdir -p "$out" /tmp/yh_sweep
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -fPIC -std=c++17 -I$R/include -I$R/yocto-hair_amd/csrc $flags \
      -c $R/yocto-hair_amd/csrc/kernels.hip -o /tmp/yh_sweep/k_$name.o &
done
wait
for v in "$@"; do
  name=${v%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/yh_sweep/libyhair_$name.so /tmp/yh_sweep/k_$name.o $(ls $R/yocto-hair_amd/csrc/*.o | grep -v kernels.o) \
      $R/yocto-hair_amd/host/*.o -lpthread -lz
  echo "== $name" | tee -a "$out/summary.txt"
  YHAIR_LIB=/tmp/yh_sweep/libyhair_$name.so python3 $R/oracle/divergence_report.py 2>&1 | tee "$out/div_$name.txt" | grep -E "ref-sphere|ref-straight|zoom|curly" | tee -a "$out/summary.txt"
done
for r in 1 2; do
  for v in "$@"; do
    name=${v%%:*}
    printf "%s r%d C1: " "$name" "$r" | tee -a "$out/summary.txt"
    YHAIR_LIB=/tmp/yh_sweep/libyhair_$name.so python3 $R/bench.py --no-cpu-baseline --steps 12 --warmup 3 2>&1 | grep -o '"value": [0-9.]*' | tr '\n' ' ' | tee -a "$out/summary.txt"
    echo | tee -a "$out/summary.txt"
  done
done
