#!/bin/bash
# round 6, GPU call 5: the upload made on the device (bounds, tree, leaf records, wide collapses): parity tests that touch it, then the command's laps
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g5; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_collapses or device_bvh or closest_hits or crowded or launch_shapes_and_kernels or ragged or light_sampling or cli_matches or mean_shift or images_match" 2>&1 | tail -15 | tee $out/pytest_subset.txt
S=$(python3 -c "
import sys; sys.path.insert(0,'tools'); import make_scenes
print(make_scenes.ensure_scene('sphere-hairblock','/tmp/yhair_scenes',scale=1.0))" | tail -1)
export YHAIR_CACHE_DIR=/tmp/yh_e2e_cache; rm -rf $YHAIR_CACHE_DIR
for k in cold warm warm2; do sleep 2
  echo "--- $k" | tee -a $out/e2e_laps.txt
  ( time YHAIR_TIMING=1 yocto-hair_amd/yscenetrace $S -r 720 -s 1536 -o /tmp/out.pfm --timing ) 2>&1 | grep -v amdgpu.ids | tee -a $out/e2e_laps.txt
done
