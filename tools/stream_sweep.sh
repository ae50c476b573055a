#!/bin/bash
# Developer tool: A/B timing of k_stream build variants inside ONE gpurun call (box-to-box spread is several per cent).
# usage: tools/stream_sweep.sh OUTDIR "scene res spp" "name:-Dflags" ...      (env: SLOTS="256 384": YHAIR_ST_SLOTS values;
#        SRC=kernels SHAPE=1: variants of csrc/kernels.hip timed with YHAIR_SHAPE=1)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/$1; cfg=$2; shift 2
mkdir -p $out /tmp/yh_sweep
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  src=$R/yocto-hair_amd/csrc
  case $name in old*) src=$R/tools/_old_csrc;; esac   # a snapshot of an earlier csrc/ placed there by hand (git-ignored)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -fPIC -std=c++17 -I$R/include -I$src $flags \
      -c $src/${SRC:-stream}.hip -o /tmp/yh_sweep/s_$name.o &
done
wait
for v in "$@"; do
  name=${v%%:*}
  others=""; for o in kernels wavefront stream; do [ "$o" != "${SRC:-stream}" ] && others="$others $R/yocto-hair_amd/csrc/$o.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/yh_sweep/libyhair_$name.so /tmp/yh_sweep/s_$name.o $others \
      $R/yocto-hair_amd/csrc/bvh_gpu.o $R/yocto-hair_amd/host/context.o $R/yocto-hair_amd/host/bvh_build.o $R/yocto-hair_amd/host/scene_io.o -lpthread -lz || exit 1
done
for P in ${SLOTS:-0}; do
  for v in "$@"; do
    name=${v%%:*}
    [ "$P" != 0 ] && export YHAIR_ST_SLOTS=$P
    YHAIR_LIB=/tmp/yh_sweep/libyhair_$name.so timeout -k 10 300 python3 $R/tools/wf_check.py $cfg ${SHAPE:-3} 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$name P=$P: /" | tee -a $out/sweep.txt
  done
done
