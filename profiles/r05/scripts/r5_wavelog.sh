cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5wavelog; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
for cfg in "straight-hair 720 64" "curly-hair 1280 32" "hair-curls 1280 32"; do
  set -- $cfg
  echo "=== $1" | tee -a gpurun_out/r5wavelog/wavelog.txt
  YHAIR_ST_WAVELOG=1 timeout -k 10 200 python3 tools/shape_check.py $1 $2 $3 3 2>&1 | grep -v amdgpu.ids | tail -24 | tee -a gpurun_out/r5wavelog/wavelog.txt
done
