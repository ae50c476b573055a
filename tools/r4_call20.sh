set -e
mkdir -p gpurun_out/r4t; O=gpurun_out/r4t
for w in 1 2 8; do
 for lib in yocto-hair_amd/libyhair.so tools/_ab/libyhair_w_nowb.so yocto-hair_amd/libyhair.so tools/_ab/libyhair_w_nowb.so; do
  echo "lib=$lib world $w"
  YHAIR_LIB=$lib timeout -k 10 200 python tools/shape_check.py sphere-hairblock 720 64 4,5,6,7,8 $w 2>&1 | grep -v "^$" | grep -v amdgpu.ids
 done
done > $O/wide_blob_ab.txt 2>&1
cat $O/wide_blob_ab.txt
