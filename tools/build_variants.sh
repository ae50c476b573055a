#!/bin/bash
# Developer tool (build container, no GPU): builds variants of ONE device source with extra -D flags and links each into
# tools/_ab/libyhair_<name>.so next to the product's other objects (git-ignored, travels to the GPU box with the snapshot),
# so that a gpurun call spends its minutes measuring, not compiling. Time them with YHAIR_LIB=tools/_ab/libyhair_<name>.so.
# usage: [SRC=stream|kernels|wide] [HOSTFLAGS="-DX"] tools/build_variants.sh "name:-DFLAG=1 -DOTHER" ...
#        HOSTFLAGS: the host sources are compiled once more with these flags (into /tmp/yh_var/host) and linked instead of host/*.o
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/yocto-hair_amd; SRC=${SRC:-stream}
mkdir -p $R/tools/_ab /tmp/yh_var
FLAGS=$(sed -n 's/^HIPFLAGS := //p' $P/Makefile | sed 's/\$(ARCH)/gfx950/; s/-I\.\.\/include//; s/-Icsrc//')
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS -I$R/include -I$P/csrc $flags -c $P/csrc/$SRC.hip -o /tmp/yh_var/${SRC}_$name.o &
done
wait
HOSTOBJ="$P/host/*.o"
if [ -n "${HOSTFLAGS:-}" ]; then
  mkdir -p /tmp/yh_var/host; CXXF=$(sed -n 's/^CXXFLAGS := //p' $P/Makefile | sed 's#\$(ROCM)#/opt/rocm#; s#-I\.\./include#-I'$R'/include#')
  for f in $P/host/context.cpp $P/host/scene_upload.cpp $P/host/launch_plan.cpp $P/host/trace_launch.cpp $P/host/gather.cpp $P/host/batch_api.cpp $P/host/bvh_build.cpp $P/host/scene_io.cpp; do
    g++ $CXXF $HOSTFLAGS -I$P/host -c $f -o /tmp/yh_var/host/$(basename ${f%.cpp}).o &
  done; wait
  HOSTOBJ="/tmp/yh_var/host/*.o"
fi
for v in "$@"; do
  name=${v%%:*}
  others=""; for o in kernels wide exact stream bvh_gpu; do [ "$o" != "$SRC" ] && others="$others $P/csrc/$o.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_ab/libyhair_$name.so /tmp/yh_var/${SRC}_$name.o $others \
      $HOSTOBJ -lpthread -lz -ldl
  echo "built tools/_ab/libyhair_$name.so"
done
