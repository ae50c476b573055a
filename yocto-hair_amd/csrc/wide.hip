// wide.hip — the sample-loop kernels that give a path MORE than four lanes (gfx950): octets over 8-wide nodes (launch shapes 4, 7),
// sixteen lanes over 16-wide nodes (6, 8), and the side-by-side launch (5: quads + the top items as octets in one kernel).
// Split from kernels.hip in round 4 (one translation unit took 77 s to compile); same templates (csrc/dev_items.h), same bits.
// Compiled with -ffp-contract=off (see dev_math.h).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "yhair.h"
#include "dev_items.h"

extern "C" {

// the kernel of a wide launch shape, or NULL when this build does not contain it
trace_kernel_t yhk_wide_kernel(int counted, int general, int shape) {
  // (instrumented builds of the 8-wide forms: plain scenes only; their per-quad counters count an octet twice, the wave-level ones hold)
  if (shape == 4 && counted && !general) return k_trace<true, false, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_OCT>;
  if (shape == 4 && !counted) return general ? k_trace<false, true, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_OCT> : k_trace<false, false, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_OCT>;
  if (shape == 6 && !counted) return general ? k_trace<false, true, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_HEX> : k_trace<false, false, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_HEX>;
  if (shape == 6 && counted && !general) return k_trace<true, false, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_HEX>;
  if (shape == 7 && !counted) return general ? k_trace<false, true, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_OCTP> : k_trace<false, false, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_OCTP>;
  if (shape == 8 && !counted) return general ? k_trace<false, true, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_HEXP> : k_trace<false, false, YH_OCT_BLOCK, YH_MIN_WAVES, YH_MODE_HEXP>;
  return nullptr;
}

// (side by side: both forms at YH_BLOCK threads; the LDS of the larger layout)
static size_t sbs_lds(const yhd_scene* sc) {
  const size_t stacks = (size_t)std::max((sc->stack_entries + YH_HITROWS) * (YH_BLOCK / 4), (sc->stack_entries8 + YH_HITROWS) * (YH_BLOCK / 8)) * 4;
  return stacks + (size_t)YHD_LDS_TABLES_F4(sc) * 16;
}
int yhk_trace_sbs_lds_bytes(const yhd_scene* sc) { return (int)sbs_lds(sc); }
int yhk_trace_sbs_occupancy(int lds_bytes, int general) {
  int  blocks = 0;
  auto k      = general ? k_trace_sbs<true> : k_trace_sbs<false>;
  if (lds_bytes > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k, YH_BLOCK, lds_bytes) != hipSuccess) return 1;
  return blocks < 1 ? 0 : blocks;
}
int yhk_trace_sbs(const yhd_scene* sc, const yhd_state* st, int nsamples, int oct_blocks, int quad_items, int oct_entries, int grid_blocks,
    hipStream_t stream) {
  const size_t lds = sbs_lds(sc);
  auto         k   = sc->general_materials ? k_trace_sbs<true> : k_trace_sbs<false>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid_blocks), dim3(YH_BLOCK), lds, stream, *sc, *st, nsamples, oct_blocks, quad_items, oct_entries);
  return (int)hipGetLastError();
}
}
