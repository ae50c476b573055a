// wavefront.hip — the persistent wavefront integrator (gfx950): trace_samples (pt.cpp:1992-2007) as
// STAGES over a pool of paths instead of one loop per pixel.
//
// k_trace (kernels.hip) runs trace -> shade -> regenerate inside one wave with one path per quad: the
// quads of a wave wait for its longest ray, and hair shading, surface shading, misses and camera rays
// of different quads execute one after the other. That is the right shape when a few expensive pixels
// bound the launch (C1). When every pixel is expensive (dense hair, BASELINE configs C2-C4) it leaves
// half of the lanes idle. Here a workgroup owns a POOL of path slots (SoA ray / hit / path state in
// HBM, yhd_pool) and alternates stages, with compaction between them:
//
//   refill    free slots take pixels from the launch's work-item list (most expensive items first)
//   trace     the rays of all live slots, as a compacted list: quads pull rays until the list is dry
//             (dev_queue.h) and sort the results by what was hit: hair list / surface list / miss
//   shade     hair list, then surface list: path_step (dev_path.h) on 16 paths of ONE kind per wave
//   finish    one LANE per slot: misses look up the environment, ended paths are clamped and
//             accumulated (trace_sample, pt.cpp:1683-1688), pixels with samples left get their next
//             camera ray, finished pixels free their slot
//
// Every pixel still has ONE path in flight and draws from its own PCG32 stream in the reference's
// order, and every stage runs the arithmetic of dev_path.h / dev_trace.h, so images are BIT-IDENTICAL
// to k_trace's (tests/test_gpu_parity.py::test_launch_shapes_and_kernels_render_identical_pixels).
#include <hip/hip_runtime.h>

#include "yhair.h"
#include "dev_path.h"
#include "dev_queue.h"

using namespace yhd;

#ifndef YH_WF_BLOCK
#define YH_WF_BLOCK 256
#endif
#ifndef YH_WF_WAVES
#define YH_WF_WAVES 6 /* waves per SIMD the register allocator must allow */
#endif

enum { C_HEAD0, C_HEAD1, C_NTRACE0, C_NTRACE1, C_NHAIR, C_NSURF, C_NREDO, C_HEADX, C_NFREE, C_GOT_BASE, C_GOT_N, C_NOMORE, C_COUNT = 16 };

YH_DEV int   path_flags(const path_t& ps) { return (ps.bounce & 255) | (ps.hit ? 256 : 0) | (ps.in_medium ? 512 : 0); }

template <bool GENERAL, int BLOCK, int WAVES, int K>
__global__ __launch_bounds__(BLOCK, WAVES) void k_wavefront(const yhd_scene sc, const yhd_state st, int nsamples, const yhd_pool pl) {
  constexpr int P = BLOCK * K, QUADS = BLOCK / 4;
  extern __shared__ v4f lds_dyn[];
  // LDS: [stacks][scene table][camera][pix left item work][six slot lists][slot state][counters]
  YH_LDS unsigned int*   lds_stack = (YH_LDS unsigned int*)lds_dyn;
  YH_LDS v4f*            lds_tabs  = (YH_LDS v4f*)(lds_stack + pl.stack_entries * QUADS);  // dev_trace.h: stage_tables
  YH_LDS int*            s_pix     = (YH_LDS int*)(lds_tabs + YHD_LDS_TABLES_F4(&sc));
  YH_LDS int*            s_left    = s_pix + P;
  YH_LDS int*            s_item    = s_left + P;
  YH_LDS unsigned int*   s_work    = (YH_LDS unsigned int*)(s_item + P);
  YH_LDS unsigned short* l_trace0  = (YH_LDS unsigned short*)(s_work + P);
  YH_LDS unsigned short* l_trace1  = l_trace0 + P;
  YH_LDS unsigned short* l_hair    = l_trace1 + P;
  YH_LDS unsigned short* l_surf    = l_hair + P;
  YH_LDS unsigned short* l_redo    = l_surf + P;
  YH_LDS unsigned short* l_free    = l_redo + P;
  YH_LDS unsigned char*  s_state   = (YH_LDS unsigned char*)(l_free + P);
  YH_LDS int*            ctr       = (YH_LDS int*)(s_state + P);

  const int    tid = threadIdx.x, q = tid & 3, quad = tid >> 2;
  const size_t base = (size_t)blockIdx.x * P;
  for (int s = tid; s < P; s += BLOCK) s_state[s] = YH_SLOT_FREE, l_free[s] = (unsigned short)s, s_work[s] = 0;
  if (tid < C_COUNT) ctr[tid] = tid == C_NFREE ? P : 0;
  trace_ctx tc;
  tc.sc = &sc, tc.lds_nodes = nullptr, tc.stats = nullptr, tc.ls = nullptr, tc.sc_dev = nullptr;
  YH_LDS float* lds_cam;
  stage_tables(sc, lds_tabs, tid, BLOCK, tc, lds_cam);
  tc.lds_stack = lds_stack + quad;
  __syncthreads();

  queue_io io;
  io.ray_o = pl.ray_o, io.ray_d = pl.ray_d, io.hit = pl.hit, io.state = s_state, io.work = s_work;
  io.hair_list = l_hair, io.surf_list = l_surf, io.redo_list = l_redo;
  io.n_hair = ctr + C_NHAIR, io.n_surf = ctr + C_NSURF, io.n_redo = ctr + C_NREDO;

  // Next camera sample of the pixel in `slot` (trace_sample, pt.cpp:1676-1682), by one lane.
  auto begin_path = [&](int slot) {
    const int p = s_pix[slot];
    rng_t     rng;
    rng.state = st.rng_state[p], rng.inc = st.rng_inc[p];
    float lu = rand1f(rng), lv = rand1f(rng);
    float pu = rand1f(rng), pv = rand1f(rng);
    st.rng_state[p] = rng.state;
    yhd_camera cam;
    for (int k = 0; k < 12; k++) cam.frame[k] = lds_cam[k];
    cam.lens = lds_cam[12], cam.film_x = lds_cam[13], cam.film_y = lds_cam[14], cam.focus = lds_cam[15], cam.aperture = lds_cam[16];
    ray_t r = sample_camera_lane(cam, p % st.width, p / st.width, st.width, st.height, pu, pv, lu, lv);
    const size_t g = base + slot;
    pl.ray_o[g]    = yhd_float4{r.o.x, r.o.y, r.o.z, 0.0f};
    pl.ray_d[g]    = yhd_float4{r.d.x, r.d.y, r.d.z, __int_as_float(0)};
    pl.weight[g]   = yhd_float4{1.0f, 1.0f, 1.0f, 0.0f};
    pl.radiance[g] = yhd_float4{0.0f, 0.0f, 0.0f, 0.0f};
    s_left[slot]--;
    s_state[slot] = YH_SLOT_RAY;
  };

  int round = 0;
  while (true) {
    const int tl = round & 1, nl = tl ^ 1;  // the trace list of this round / the one being filled for the next
    YH_LDS unsigned short* l_tl = tl ? l_trace1 : l_trace0;
    YH_LDS unsigned short* l_nl = tl ? l_trace0 : l_trace1;
    // ---- refill: free slots take the pixels of the next work items ---------------------------------
    if (tid == 0) {
      ctr[C_NTRACE0 + nl] = 0, ctr[C_HEAD0 + nl] = 0, ctr[C_NHAIR] = 0, ctr[C_NSURF] = 0, ctr[C_NREDO] = 0, ctr[C_HEADX] = 0;
      int nfree = ctr[C_NFREE], want = ctr[C_NOMORE] ? 0 : nfree / 16, t0 = 0, got = 0;
      if (want > 0) {
        t0  = atomicAdd(st.tile_cursor, want);
        got = max(0, min(want, st.num_tiles - t0));
        if (got < want) ctr[C_NOMORE] = 1;
      }
      ctr[C_GOT_BASE] = t0, ctr[C_GOT_N] = got, ctr[C_NFREE] = nfree - 16 * got;
    }
    __syncthreads();
    const int  got = ctr[C_GOT_N], t0 = ctr[C_GOT_BASE], nf_after = ctr[C_NFREE];
    const bool nomore = ctr[C_NOMORE] != 0;
    auto item_pixel = [&](int r, int& item, int& pixel) {  // pixel r & 15 of the r / 16-th item taken
      item     = st.tiles[t0 + (r >> 4)];
      int tile = item >> 2, part = item & 3, pq = r & 15;
      int i    = (tile % st.tiles_x) * YH_TILE + (part & 1) * 4 + (pq & 3);
      int j    = (tile / st.tiles_x) * YH_TILE + (part >> 1) * 4 + (pq >> 2);
      pixel    = (i < st.width && j < st.height) ? j * st.width + i : -1;
    };
    int beyond[K];  // slots this thread took for pixels beyond the image edge (-1: none)
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int r = tid + k * BLOCK;
      beyond[k]   = -1;
      if (r < 16 * got) {
        int slot = l_free[nf_after + r], item, pixel;
        item_pixel(r, item, pixel);
        if (pixel >= 0) {
          s_pix[slot] = pixel, s_left[slot] = nsamples, s_item[slot] = item, s_work[slot] = 0;
          begin_path(slot);
        } else {
          beyond[k] = slot;
        }
        wave_append(l_tl, ctr + C_NTRACE0 + tl, pixel >= 0, slot);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++)  // those slots stay free (all reads of the popped entries happened before the barrier)
      if (beyond[k] >= 0) l_free[lds_add(ctr + C_NFREE, 1)] = (unsigned short)beyond[k];
    const int ntrace = ctr[C_NTRACE0 + tl];
    if (ntrace == 0) {
      if (nomore) break;  // no path in flight and no work item left
      __syncthreads();
      continue;           // (the items taken lay beyond the image edge: take more)
    }
    // ---- trace: all live paths' rays as one compacted list ------------------------------------------
    trace_queue<QUADS, false>(tc, io, base, l_tl, ntrace, ctr + C_HEAD0 + tl);
    __syncthreads();
    const int nredo = ctr[C_NREDO];
    if (nredo > 0) {  // axis-parallel rays: the reference's compare-and-select box test (dev_trace.h)
      trace_queue<QUADS, true>(tc, io, base, l_redo, nredo, ctr + C_HEADX);
      __syncthreads();
    }
    // ---- shade: hair hits, then surface hits; one kind per pass, one path per quad -------------------
    const int nhair = ctr[C_NHAIR], nsurf = ctr[C_NSURF];
    for (int pass = 0; pass < 2; pass++) {
      const YH_LDS unsigned short* list = pass ? l_surf : l_hair;
      const int                    n    = pass ? nsurf : nhair;
      for (int idx = quad; idx < n; idx += QUADS) {
        const int    slot = list[idx], p = s_pix[slot];
        const size_t g    = base + slot;
        yhd_float4 o = pl.ray_o[g], d = pl.ray_d[g], w = pl.weight[g], rad = pl.radiance[g];
        yhd_int4   h = pl.hit[g];
        path_t ps;
        ps.ray      = ray_t{f3{o.x, o.y, o.z}, f3{d.x, d.y, d.z}, ray_eps, flt_max};
        ps.weight   = f3{w.x, w.y, w.z}, ps.radiance = f3{rad.x, rad.y, rad.z};
        const int fl = __float_as_int(d.w);
        ps.bounce = fl & 255, ps.hit = (fl & 256) != 0, ps.in_medium = (fl & 512) != 0;
        if (GENERAL && ps.in_medium) {
          yhd_float4 m0 = pl.medium[2 * g], m1 = pl.medium[2 * g + 1];
          ps.medium.density = f3{m0.x, m0.y, m0.z}, ps.medium.anisotropy = m0.w, ps.medium.scatter = f3{m1.x, m1.y, m1.z};
        }
        hit_t isec;
        isec.object = h.x, isec.slot = h.y, isec.u = __int_as_float(h.z), isec.v = __int_as_float(h.w), isec.distance = o.w;
        rng_t rng;
        rng.state = st.rng_state[p], rng.inc = st.rng_inc[p];
        const bool alive = path_step<false, QUADS, GENERAL>(tc, ps, isec, rng, st.bounces);
        if (q == 0) {
          st.rng_state[p] = rng.state;
          pl.radiance[g]  = yhd_float4{ps.radiance.x, ps.radiance.y, ps.radiance.z, 0.0f};
          pl.ray_d[g]     = yhd_float4{ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, __int_as_float(path_flags(ps))};
          if (alive) {
            pl.ray_o[g]  = yhd_float4{ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, 0.0f};
            pl.weight[g] = yhd_float4{ps.weight.x, ps.weight.y, ps.weight.z, 0.0f};
            if (GENERAL && ps.in_medium) {
              pl.medium[2 * g]     = yhd_float4{ps.medium.density.x, ps.medium.density.y, ps.medium.density.z, ps.medium.anisotropy};
              pl.medium[2 * g + 1] = yhd_float4{ps.medium.scatter.x, ps.medium.scatter.y, ps.medium.scatter.z, 0.0f};
            }
          }
          s_state[slot] = alive ? YH_SLOT_RAY : YH_SLOT_ENDED;
        }
        wave_append(l_nl, ctr + C_NTRACE0 + nl, alive && q == 0, slot);
      }
    }
    __syncthreads();
    // ---- finish: misses, ended paths, next samples; one lane per slot -------------------------------
    for (int slot = tid; slot < P; slot += BLOCK) {
      const unsigned char s = s_state[slot];
      bool next = false;
      if (s == YH_SLOT_MISS || s == YH_SLOT_ENDED) {
        const size_t g = base + slot;
        const int    p = s_pix[slot];
        yhd_float4 rad = pl.radiance[g], d = pl.ray_d[g];
        path_t ps;
        ps.radiance = f3{rad.x, rad.y, rad.z};
        ps.hit      = (__float_as_int(d.w) & 256) != 0;
        if (s == YH_SLOT_MISS) {  // pt.cpp:1397-1400
          yhd_float4 w = pl.weight[g];
          ps.radiance = ps.radiance + f3{w.x, w.y, w.z} * eval_environment<false>(tc, f3{d.x, d.y, d.z});
        }
        yhd_float4 acc = st.accum[p];
        path_end(ps, st.clamp, acc);
        st.accum[p] = acc;
        if (s_left[slot] > 0) {
          begin_path(slot);
          next = true;
        } else {  // the pixel has all its samples: report its work, free the slot
          if (s_work[slot]) atomicAdd(&st.tile_cost[s_item[slot]], s_work[slot]);
          s_state[slot]                     = YH_SLOT_FREE;
          l_free[lds_add(ctr + C_NFREE, 1)] = (unsigned short)slot;
        }
      }
      wave_append(l_nl, ctr + C_NTRACE0 + nl, next, slot);
    }
    round++;
    __syncthreads();
  }
}

extern "C" {

typedef void (*wavefront_kernel_t)(const yhd_scene, const yhd_state, int, const yhd_pool);
static wavefront_kernel_t wavefront_kernel(bool general, int k) {
  if (k >= 2) return general ? k_wavefront<true, YH_WF_BLOCK, YH_WF_WAVES, 2> : k_wavefront<false, YH_WF_BLOCK, YH_WF_WAVES, 2>;
  return general ? k_wavefront<true, YH_WF_BLOCK, YH_WF_WAVES, 1> : k_wavefront<false, YH_WF_BLOCK, YH_WF_WAVES, 1>;
}
int yhk_wavefront_block_threads(void) { return YH_WF_BLOCK; }
int yhk_wavefront_slots(int k) { return YH_WF_BLOCK * (k >= 2 ? 2 : 1); }
int yhk_wavefront_lds_bytes(int stack_entries, int tables_f4, int k) {
  int P = yhk_wavefront_slots(k);
  return stack_entries * (YH_WF_BLOCK / 4) * 4 + tables_f4 * 16 + P * (4 * 4 + 6 * 2 + 1) + C_COUNT * 4;
}
int yhk_wavefront_occupancy(int lds_bytes, int general, int k) {
  int                blocks = 0;
  wavefront_kernel_t kern   = wavefront_kernel(general != 0, k);
  if (lds_bytes > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, YH_WF_BLOCK, lds_bytes) != hipSuccess) return 0;
  return blocks;
}
int yhk_wavefront(const yhd_scene* sc, const yhd_state* st, int nsamples, const yhd_pool* pl, int k, int grid_blocks,
    hipStream_t stream) {
  int                lds  = yhk_wavefront_lds_bytes(pl->stack_entries, YHD_LDS_TABLES_F4(sc), k);
  wavefront_kernel_t kern = wavefront_kernel(sc->general_materials != 0, k);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(grid_blocks), dim3(YH_WF_BLOCK), lds, stream, *sc, *st, nsamples, *pl);
  return (int)hipGetLastError();
}
}
