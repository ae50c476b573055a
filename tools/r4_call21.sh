set -e
mkdir -p gpurun_out/r4u; O=gpurun_out/r4u
for rep in 1 2; do
for v in s_base s_nt1 s_nt2 s_nt3 s_ntleaf s_ntall; do
  echo "lib=$v"
  YHAIR_LIB=tools/_ab/libyhair_$v.so timeout -k 10 120 python tools/shape_check.py curly-hair 1280 32 3 2>&1 | grep "shape 3" | tail -1
  YHAIR_LIB=tools/_ab/libyhair_$v.so timeout -k 10 120 python tools/shape_check.py straight-hair 720 64 3 2>&1 | grep "shape 3" | tail -1
done; done > $O/nt_ab.txt 2>&1
cat $O/nt_ab.txt
