// exact.hip — the quad sample-loop kernel with the hair BSDF's EXACT arithmetic (yh_trace_params::hair_exact).
//
// dev_hair.h evaluates the BSDF inside the stated 1e-4 tolerance with the hardware's 1-ulp reciprocal / square root /
// log2 / sin / cos and the float asinf (YH_HAIR_FAST = 1, the default of every other translation unit). This unit
// compiles the SAME sample loop (dev_items.h) with YH_HAIR_FAST = 0: IEEE divisions and square roots, the library's
// log / sin / cos and the reference's double asin (ext.cpp:111,148-151) — the arithmetic whose paths FOLLOW the
// reference's longest (relRMSE against the reference 0.23 of the seed-to-seed floor on the reference's own
// sphere-hairblock scene against 0.52, profiles/r02/bsdf_arithmetic_variants.txt) at 0.88 x the speed.
// Device functions are inline and per translation unit, so the two arithmetics never mix; the kernel has its own name.
#include <hip/hip_runtime.h>

#define YH_HAIR_FAST 0
#include "yhair.h"
#include "dev_items.h"

template <bool GENERAL>
__global__ __launch_bounds__(YH_BLOCK, YH_MIN_WAVES) void k_trace_exact(const yhd_scene sc, const yhd_state st,
    int nsamples, yhd_counters* counters) {
  trace_items<false, GENERAL, YH_BLOCK, YH_SHADER_PATH>(sc, st, nsamples, counters);
}

extern "C" {
// the launch geometry is k_trace's 512 x 4 shape (yhk_trace_lds_bytes(sc, 0), yhk_block_threads(0))
int yhk_trace_exact(const yhd_scene* sc, const yhd_state* st, int nsamples, int lds_bytes, int grid_blocks, hipStream_t stream) {
  auto k = sc->general_materials ? k_trace_exact<true> : k_trace_exact<false>;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid_blocks), dim3(YH_BLOCK), (size_t)lds_bytes, stream, *sc, *st, nsamples, (yhd_counters*)nullptr);
  return (int)hipGetLastError();
}
int yhk_trace_exact_occupancy(int lds_bytes, int general) {
  int  blocks = 0;
  auto k      = general ? k_trace_exact<true> : k_trace_exact<false>;
  if (lds_bytes > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k, YH_BLOCK, lds_bytes) != hipSuccess) return 1;
  return blocks < 1 ? 0 : blocks;
}
}
