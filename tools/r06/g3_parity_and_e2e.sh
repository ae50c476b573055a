#!/bin/bash
# round 6, GPU call 3: (a) the whole command by stage (YHAIR_TIMING laps of upload / init_state / wide nodes / requests), cold and warm trial record;
# (b) parity against the reference at the dense configs' OWN sample counts on a reduced image, with the mean-shift (bias) estimator
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g3; mkdir -p $out; export TMPDIR=/tmp
python3 -c "
import sys; sys.path.insert(0,'tools'); import make_scenes
print(make_scenes.ensure_scene('sphere-hairblock','/tmp/yhair_scenes',scale=1.0))" > $out/scene.txt
S=$(tail -1 $out/scene.txt)
export YHAIR_CACHE_DIR=/tmp/yh_e2e_cache; rm -rf $YHAIR_CACHE_DIR
for k in cold warm warm2; do
  echo "--- $k" | tee -a $out/e2e_laps.txt
  ( time YHAIR_TIMING=1 yocto-hair_amd/yscenetrace $S -r 720 -s 1536 -o /tmp/out.pfm --timing ) 2>&1 | grep -v amdgpu.ids | tee -a $out/e2e_laps.txt
done
unset YHAIR_CACHE_DIR
timeout -k 10 300 python3 tools/parity_vs_spp.py C2 180 > $out/parity_vs_spp_C2.json 2> $out/C2.err || { tail -3 $out/C2.err; echo "C2 FAILED"; }
tail -4 $out/C2.err
timeout -k 10 300 python3 tools/parity_vs_spp.py C4 256 > $out/parity_vs_spp_C4.json 2> $out/C4.err || { tail -3 $out/C4.err; echo "C4 FAILED"; }
tail -4 $out/C4.err
timeout -k 10 900 python3 tools/parity_vs_spp.py C3 256 > $out/parity_vs_spp_C3.json 2> $out/C3.err || { tail -3 $out/C3.err; echo "C3 FAILED"; }
tail -4 $out/C3.err
