// dev_items.h — the persistent sample loop of the quad kernels (k_trace, k_trace_shader, k_trace_exact): included by
// csrc/kernels.hip and by csrc/exact.hip (the same loop compiled with the exact hair-BSDF arithmetic, YH_HAIR_FAST=0).
#ifndef YH_DEV_ITEMS_H_
#define YH_DEV_ITEMS_H_
#include "dev_path.h"

using namespace yhd;

#define YH_BLOCK 512
#define YH_MIN_WAVES 4 /* waves per SIMD the register allocator must allow */

// ---------------------------------------------------------------------------
// The sample loop
// ---------------------------------------------------------------------------
// Persistent wavefronts, quads and path regeneration. A wave pulls a work item
// — one 4x4-pixel quadrant of an 8x8 tile — from the launch's queue (items are
// queued most-expensive-first using the cost each reported in the previous
// launch). Each QUAD (four adjacent lanes) owns one pixel and keeps ONE path in
// flight: the four lanes run the same path redundantly and split the work
// inside BVH steps (one box / one primitive per lane) and inside the hair BSDF
// (one lobe per lane). Every iteration traces the current segment of all live
// quads and shades it; a quad whose path ended starts its pixel's next sample
// in the same iteration (a pixel's PCG32 stream is sequential, pt.cpp:1942-1945,
// so its samples cannot run side by side) — quads never wait for the longest
// path of a sample, only for the item's last quad.
#define YH_QUADS (YH_BLOCK / 4)
// GENERAL = the scene has materials with lobes beyond diffuse / hair (dev_surface.h); scenes
// without them (all BASELINE configs) run the variant that does not carry that code.
// BLOCK x WAVES = the launch shape: 512 threads at 4 waves per SIMD (128 VGPRs) when the launch is
// bound by a few expensive pixels (C1), 256 threads at 5 waves per SIMD (96 VGPRs, more latency hiding)
// when many pixels are expensive (dense hair: +5-10 %, profiles/r01, r02); the host picks by measurement.
// SHADER = the reference's shader_type (YH_SHADER_*): trace_path is the product path (k_trace), the
// preview / debug shaders (naive, eyelight, normal) share everything but the bounce step (k_trace_shader).
// MODE (dev_trace.h): YH_MODE_QUAD = a quad per path as above, over 4-wide nodes;
// YH_MODE_OCT = EIGHT lanes per path (two quads that run the same path and share the box tests of an 8-wide node): a
// wave holds 8 pixels, a work-list entry is HALF a quadrant (entry = item << 1 | half: rows 2 half, 2 half + 1 of the
// 4x4 block) — half the paths per wave, twice the waves, for launches bound by the chain of one path.
// `block`, `blocks`: this workgroup's index among the workgroups that share `st`'s work list, and their number (the whole grid,
// except in k_trace_sbs, where two lists share one launch).
template <bool COUNT, bool GENERAL, int BLOCK, int SHADER, int MODE = YH_MODE_QUAD>
YH_DEV void trace_items(const yhd_scene& sc, const yhd_state& st, int nsamples, yhd_counters* counters, unsigned int block = blockIdx.x,
    unsigned int blocks = gridDim.x) {
  constexpr int LPP    = YH_IS_HEX(MODE) ? 16 : YH_IS_OCT(MODE) ? 8 : 4;  // lanes per path
  constexpr int GROUPS = BLOCK / LPP;                   // paths per block = columns of the LDS stack
  extern __shared__ v4f lds_dyn[];
  // LDS carve-out: [stacks: stack entries x GROUPS uint][tables: scene level | camera | small area lights | environment cdf index | materials]
  // (dev_trace.h: stage_tables)
  YH_LDS unsigned int* lds_stack = (YH_LDS unsigned int*)lds_dyn;
  YH_LDS v4f*          lds_tabs  = (YH_LDS v4f*)(lds_stack + ((MODE == YH_MODE_QUAD ? sc.stack_entries : YH_IS_HEX(MODE) ? sc.stack_entries16 : sc.stack_entries8) + YH_HITROWS) * GROUPS);
  trace_ctx tc;
  tc.sc = &sc;
  tc.ls = nullptr, tc.sc_dev = nullptr;
  YH_LDS float* lds_cam;
  stage_tables(sc, lds_tabs, threadIdx.x, blockDim.x, tc, lds_cam);
  __syncthreads();

  tc.lds_stack = lds_stack + (threadIdx.x / LPP);
  stats_t stats = {};
  tc.stats = COUNT ? &stats : nullptr;

  const int lane = threadIdx.x & 63;
  // The FIRST item of a wave is the list entry at the wave's own position (workgroup x waves per workgroup + wave), the
  // others come from the cursor, which starts behind those positions. The four wave slots of a SIMD do not run at the same
  // speed (slot 0: 10.4 ms for the kind of item that takes 13.3 ms in slot 3 on C1, profiles/r03/where_items_ran.txt) and a
  // wave's slot follows from the dispatch order of its workgroup, so the host lays the head of the list out by position: the
  // most expensive items on the fastest slots, none on the slowest (host/launch_plan.cpp: lay_out_first_round). Any list is
  // rendered correctly — every entry is taken exactly once, by position or through the cursor; the layout is a matter of time.
  const int by_position = (int)(blocks * (BLOCK / 64));
  int       t_first     = __builtin_amdgcn_readfirstlane((int)(block * (BLOCK / 64) + (threadIdx.x >> 6)));
  while (true) {
    int t = t_first;
    if (t < 0) {
      if (lane == 0) t = atomicAdd(st.tile_cursor, 1);
      t = __builtin_amdgcn_readfirstlane(t) + by_position;
    }
    t_first = -1;
    if (t >= st.num_tiles) break;
    unsigned long long t0 = wall_clock64();
    int  item  = st.tiles[t];
    int  half  = 0;  // which half (octets) / quarter (sixteen lanes per path) of the quadrant this entry is
    if (YH_IS_OCT(MODE)) half = item & 1, item >>= 1;
    if (YH_IS_HEX(MODE)) half = item & 3, item >>= 2;
    int  tile  = item >> 2, part = item & 3;
    int  pq    = YH_IS_HEX(MODE) ? half * 4 + (lane >> 4) : YH_IS_OCT(MODE) ? half * 8 + (lane >> 3) : lane >> 2;  // pixel of the 4x4 quadrant owned by this lane's group
    int  i     = (tile % st.tiles_x) * YH_TILE + (part & 1) * 4 + (pq & 3);
    int  j     = (tile / st.tiles_x) * YH_TILE + (part >> 1) * 4 + (pq >> 2);
    bool owner = i < st.width && j < st.height;
    size_t pix = owner ? (size_t)j * st.width + i : 0;
    rng_t  rng;
    rng.state      = st.rng_state[pix];
    rng.inc        = st.rng_inc[pix];
    yhd_float4 acc = st.accum[pix];
    int    left    = owner ? nsamples : 0;  // samples this quad still has to start
    bool   alive   = false;
    path_t ps;
    ps.bounce = 0, ps.hit = false;
    unsigned long long cyc_trace = 0, cyc_shade = 0;
    unsigned int       w_iters = 0, w_steps = 0, l_steps = 0, l_iters = 0;
    while (true) {
      if (!alive && left > 0) {
        yhd_camera cam;
        for (int k = 0; k < 12; k++) cam.frame[k] = lds_cam[k];
        cam.lens = lds_cam[12], cam.film_x = lds_cam[13], cam.film_y = lds_cam[14], cam.focus = lds_cam[15], cam.aperture = lds_cam[16];
        path_begin(cam, ps, rng, i, j, st.width, st.height);
        left--;
        alive = true;
      }
      if (!__any(alive)) break;
      unsigned long long c0 = 0, c1 = 0;
      unsigned int       steps = 0;
      if (COUNT) c0 = clock64(), l_iters += (alive && (lane & 3) == 0) ? 1 : 0, w_iters++;
      hit_t isec;
      if (alive) {
        if (COUNT) count_quad<COUNT>(stats.rays);
        isec = trace_ray<COUNT, GROUPS, !GENERAL, MODE>(tc, ps.ray, -1, &steps);
      }
      if (COUNT) {
        c1 = clock64(), cyc_trace += c1 - c0;
        unsigned int smax = steps;
        for (int off = 32; off > 0; off >>= 1) smax = max(smax, (unsigned int)__shfl_xor((int)smax, off, 64));
        w_steps += smax, l_steps += (lane & 3) == 0 ? steps : 0;
      }
      if (alive) {
        if constexpr (SHADER == YH_SHADER_PATH) alive = path_step<COUNT, GROUPS, GENERAL>(tc, ps, isec, rng, st.bounces);
        else alive = shade_step<COUNT, GROUPS, SHADER>(tc, ps, isec, rng, st.bounces);
        if (!alive) {
          path_end(ps, st.clamp, acc);
          if (COUNT) count_quad<COUNT>(stats.samples);
        }
      }
      if (COUNT) cyc_shade += clock64() - c1;
    }
    if (COUNT) {  // flush this item's counters: one wave reduction, one atomic per counter
      unsigned int v[11] = {stats.samples, stats.rays, stats.nodes, stats.seg, stats.tri, stats.hair, stats.surf,
          stats.envl, stats.envs, l_steps, l_iters};
      for (int k = 0; k < 11; k++)
        for (int off = 32; off > 0; off >>= 1) v[k] += (unsigned int)__shfl_xor((int)v[k], off, 64);
      if (lane == 0) {
        atomicAdd(&counters->samples, (unsigned long long)v[0]), atomicAdd(&counters->rays, (unsigned long long)v[1]);
        atomicAdd(&counters->nodes, (unsigned long long)v[2]), atomicAdd(&counters->seg, (unsigned long long)v[3]);
        atomicAdd(&counters->tri, (unsigned long long)v[4]), atomicAdd(&counters->hair, (unsigned long long)v[5]);
        atomicAdd(&counters->surf, (unsigned long long)v[6]), atomicAdd(&counters->envl, (unsigned long long)v[7]);
        atomicAdd(&counters->envs, (unsigned long long)v[8]);
        atomicAdd(&counters->lane_steps, (unsigned long long)v[9]), atomicAdd(&counters->lane_iters, (unsigned long long)v[10]);
        atomicAdd(&counters->cyc_trace, cyc_trace), atomicAdd(&counters->cyc_shade, cyc_shade);
        atomicAdd(&counters->wave_iters, (unsigned long long)w_iters), atomicAdd(&counters->wave_steps, (unsigned long long)w_steps);
      }
      if (lane == 0) {
        atomicAdd(&counters->c_geom, stats.c_geom), atomicAdd(&counters->c_sample, stats.c_sample);
        atomicAdd(&counters->c_eval, stats.c_eval), atomicAdd(&counters->c_rest, stats.c_rest);
      }
      {
        unsigned int b[10] = {stats.t_node, stats.l_node, stats.t_line, stats.l_line, stats.t_tri, stats.l_tri, stats.t_enter,
            stats.l_enter, stats.t_scene, stats.l_scene};
        for (int k = 0; k < 10; k++) {
          for (int off = 32; off > 0; off >>= 1) b[k] += (unsigned int)__shfl_xor((int)b[k], off, 64);
          if (lane == 0) atomicAdd(&counters->branch[k], (unsigned long long)b[k]);
        }
      }
      stats = stats_t{};
    }
    if (owner && (lane & (LPP - 1)) == 0) {
      st.rng_state[pix] = rng.state;
      st.accum[pix]     = acc;
    }
    if (lane == 0) {
      unsigned int dt = (unsigned int)(wall_clock64() - t0);
      if (YH_IS_OCT(MODE) || YH_IS_HEX(MODE)) atomicAdd(&st.tile_cost[item], dt);  // the halves / quarters of a quadrant add up (zeroed before the launch)
      else st.tile_cost[item] = dt;
      if (COUNT) atomicAdd(&counters->cyc_tile, (unsigned long long)dt);
    }
  }
}
// ---------------------------------------------------------------------------
// The sample-loop kernels (instantiated by csrc/kernels.hip: quads over 4-wide nodes, the other shaders; csrc/wide.hip: octets,
// sixteen lanes per path, side by side; csrc/exact.hip has its own)
// ---------------------------------------------------------------------------
typedef void (*trace_kernel_t)(const yhd_scene, const yhd_state, int, yhd_counters*);
#define YH_OCT_BLOCK 256 /* workgroup of the wide forms (launch shapes 4, 6, 7, 8) */
template <bool COUNT, bool GENERAL, int BLOCK, int WAVES, int MODE = YH_MODE_QUAD>
__global__ __launch_bounds__(BLOCK, WAVES) void k_trace(const yhd_scene sc, const yhd_state st,
    int nsamples, yhd_counters* counters) {
  trace_items<COUNT, GENERAL, BLOCK, YH_SHADER_PATH, MODE>(sc, st, nsamples, counters);
}
// SIDE BY SIDE in one launch (launch shape 5): the first `oct_blocks` workgroups run the octet form over the second part of
// the work list (`oct_entries` half-quadrant entries behind the `quad_items` quad entries, its own cursor), the others the
// quad form over the first part. The first workgroups of a launch get the fastest wave slots of their CUs (dev_items.h), so
// the few items whose chain bounds the launch run with eight lanes per path AND in the best slots; same workgroup size, so
// the two forms pack on a CU like one kernel's workgroups.
template <bool GENERAL>
__global__ __launch_bounds__(YH_BLOCK, YH_MIN_WAVES) void k_trace_sbs(const yhd_scene sc, const yhd_state st, int nsamples, int oct_blocks,
    int quad_items, int oct_entries) {
  if ((int)blockIdx.x < oct_blocks) {
    yhd_state so   = st;
    so.tiles       = st.tiles + quad_items, so.num_tiles = oct_entries, so.tile_cursor = st.tile_cursor + 16;
    trace_items<false, GENERAL, YH_BLOCK, YH_SHADER_PATH, YH_MODE_OCT>(sc, so, nsamples, nullptr, blockIdx.x, (unsigned)oct_blocks);
  } else {
    yhd_state sq = st;
    sq.num_tiles = quad_items;
    trace_items<false, GENERAL, YH_BLOCK, YH_SHADER_PATH, YH_MODE_QUAD>(sc, sq, nsamples, nullptr, blockIdx.x - (unsigned)oct_blocks, gridDim.x - (unsigned)oct_blocks);
  }
}
#endif
