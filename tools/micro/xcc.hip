// xcc.hip — which XCD does block b run on?  (HW_REG_XCC_ID vs blockIdx % 8)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20);
}
int main() {
  int n = 4096, *d, h[4096];
  hipMalloc(&d, n * 4);
  k<<<n, 512>>>(d);
  hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  int hist[16] = {0}, match = 0;
  for (int i = 0; i < n; i++) hist[h[i] & 15]++, match += ((h[i] & 7) == (i & 7));
  for (int i = 0; i < 16; i++) printf("xcc %d: %d\n", i, hist[i]);
  printf("first 16:"); for (int i = 0; i < 16; i++) printf(" %d", h[i]); printf("\nmatch blockIdx%%8: %d / %d\n", match, n);
  return 0;
}
