// Developer micro-benchmark: dependent-load latency seen by ONE wave on MI355X for
// working sets of different sizes (hipMalloc memory), 16-byte loads like the BVH fetches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const uint4* buf, int steps, unsigned start, unsigned long long* out, unsigned* sink) {
  unsigned idx = start + threadIdx.x;  // each lane its own chain
  unsigned long long t0 = wall_clock64();
  unsigned long long c0 = clock64();
  for (int i = 0; i < steps; i++) idx = buf[idx].x;
  unsigned long long c1 = clock64();
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = c1 - c0;
  sink[threadIdx.x] = idx;
}
int main() {
  for (size_t mb : {1, 16, 64, 256, 1024}) {
    size_t n = mb * 1024 * 1024 / 16;
    std::vector<unsigned> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<uint4> h(n);
    for (size_t i = 0; i < n; i++) h[perm[i]].x = perm[(i + 1) % n];  // one big random cycle
    uint4* d; unsigned long long* out; unsigned* sink;
    hipMalloc(&d, n * 16); hipMalloc(&out, 16); hipMalloc(&sink, 256);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
      int steps = 2000;
      hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, steps, 0u, out, sink);
      unsigned long long o[2];
      hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
      printf("%5zu MB rep %d: %.1f ns per dependent 16-B load (wall), %.0f shader clocks per load, clock %.2f GHz\n", mb, rep,
             o[0] * 10.0 / steps, (double)o[1] / steps, (double)o[1] / (o[0] * 10.0));
    }
    hipFree(d); hipFree(out); hipFree(sink);
  }
  return 0;
}
