// stream.hip — the streaming integrator (gfx950): trace_samples (pt.cpp:1992-2007) with ONE LANE PER PATH
// and a pool of paths per WAVEFRONT, for scenes where every pixel is expensive (dense hair:
// BASELINE configs C2-C4).
//
// k_trace (kernels.hip) gives a path four lanes and a whole loop iteration; that is the right shape
// when a few expensive pixels bound the launch (C1) and wasteful when there are more expensive pixels
// than lanes: the counters of C2 show 28 % of the vector lanes active, everything outside box /
// primitive / lobe steps computed four times, and a wave waiting for its longest ray every iteration.
// Here a wavefront owns `slots_per_wave` path slots (SoA in HBM, yhd_stream) and moves them through
// STAGES, 64 paths of one kind at a time, with the lists between the stages in its own LDS:
//
//   items    free slots take pixels from the launch's work-item list
//   finish   ended samples are clamped and accumulated (trace_sample, pt.cpp:1683-1688), misses look up
//            the environment first; pixels with samples left get their next camera ray
//   trace    one ray per lane (dev_lane.h); a lane that finishes its ray takes the next one of the ray
//            list; when the list is dry and at most yhd_stream::suspend_lanes lanes are busy the wave goes
//            shading and the unfinished rays stay where they are — in registers, their stacks in the
//            lanes' LDS columns — until the next trace stage
//   sort     finished rays by what they hit: hair / surface / miss
//   shade    path_step (dev_path.h) on 64 hair hits, or 64 surface hits; partial batches only when the
//            wave has nothing else to fill its lanes with
//
// A wave synchronises with nobody: no workgroup barrier after the scene table is staged, no atomics but
// the work-item cursor. Every pixel still has ONE path in flight and draws from its own PCG32 stream in
// the reference's order, and every stage runs the arithmetic of dev_path.h / dev_trace.h in its
// one-lane forms (YH_LANE), so images are BIT-IDENTICAL to k_trace's
// (tests/test_gpu_parity.py::test_launch_shapes_and_kernels_render_identical_pixels).
#define YH_LANE 1
#include <hip/hip_runtime.h>

#include "yhair.h"
#include "dev_path.h"

using namespace yhd;

#define YH_ST_BLOCK 256
#define YH_ST_WAVES 4        /* waves per SIMD the register allocator must allow (5 = 96 registers puts 28 spill instructions into the step loop: profiles/r05/coop_line_leaves.txt) */
#ifndef YH_ST_WAVES_GENERAL
#define YH_ST_WAVES_GENERAL 4 /* ... of the GENERAL variant (surface lobes, volumes, textures, big lights) */
#endif
#define YH_ST_WAVE_LDS(P) (64 * YH_LSTACK * 4 + 64 * 8 + 6 * (P) * 2) /* LDS of one wave */
#define YH_REFILL_LANES 16   /* idle lanes of a wave before the (divergent) refill code runs */
#define YH_SUSPEND_LANES 16  /* ray list dry and at most this many lanes busy: go shading. The default of yhd_stream::suspend_lanes, which the host sets:
                                16 for images of several generations, 8 when the resident pools hold the whole image at once (C2: 303-316 -> 330-339 Msamples/s
                                with the cooperative leaves, C3 349 -> 334-340 the other way: profiles/r05/coop_line_leaves.txt) */
// (tuned in rounds 2 and 4, profiles/r02/k_stream_sweep_slots_suspend.txt, profiles/r04/k_stream_tuning_after_blob.txt; the other
// scheduling policy, items taken sixteen slots at a time and the field-by-field pool layout are closed A/Bs of the same records)

// Fields of a path slot (yh_device.h: yhd_path_slot): eight 16-byte fields in one 128-byte line, GROUPED BY WHO WRITES THEM into the line's four 32-byte
// sectors (round 6) — a stage that retires a ray dirties one sector, a shaded bounce two, not the whole line: the pool is 20 x the L2s, so a line is
// written back between two stages that touch it, and what leaves the L2 is its dirty sectors.
//   sector 0   ray origin | ray direction                                   written by: finish (a new camera ray), shade (the next ray of a live path)
//   sector 1   weight.xyz, flags | rng state lo, hi, traversal steps so far  ...        finish, shade (every bounce)
//   sector 2   hit: object, leaf slot, u, v | distance, steps of this ray    ...        the trace stage, when it retires the ray (publish); shade: object = H_ENDED
//   sector 3   radiance.xyz, pixel | samples left, work item, rng inc lo, hi ...        items (a new pixel), finish (a new sample), shade only when the bounce met an emitter
#define SLOT_F4(pl, g, k) (((yhd_float4*)&(pl).slots[g])[k])
#define SLOT_I4(pl, g, k) (((yhd_int4*)&(pl).slots[g])[k])
#define SLOT_RAY_O(pl, g) SLOT_F4(pl, g, 0)
#define SLOT_RAY_D(pl, g) SLOT_F4(pl, g, 1)
#define SLOT_WEIGHT(pl, g) SLOT_F4(pl, g, 2) /* .w = int bits: path_flags */
#define SLOT_RNGW(pl, g) SLOT_I4(pl, g, 3)   /* rng state lo, hi, traversal steps of the pixel's rays so far (a scheduling hint), - */
#define SLOT_HIT(pl, g) SLOT_I4(pl, g, 4)
#define SLOT_HIT2(pl, g) SLOT_I4(pl, g, 5)   /* distance (float bits), steps of the ray just traced, -, - */
#define SLOT_RADIANCE(pl, g) SLOT_F4(pl, g, 6) /* .w = int bits: the pixel */
#define SLOT_OWN(pl, g) SLOT_I4(pl, g, 7)    /* samples left to start, work item, rng inc lo, hi */

// Progress: a trace stage that leaves with rays suspended (<= suspend_lanes busy lanes, something pending) must be
// followed by a stage that consumes what is pending — the scheduler flushes partial batches when fewer than 64 rays are at
// hand — or the two would hand the wave back and forth forever.
static_assert(YH_SUSPEND_LANES < 64, "the trace stage's exit condition must imply the scheduler's flush condition");  // (yhk_stream refuses suspend_lanes >= 64)
enum { K_HAIR = 0, K_SURF = 1, K_MISS = 2, K_REDO = 3 };  // what a finished ray found (bits 12-13 of a done-list entry)
enum { H_MISS = -1, H_ENDED = -2, H_NEW = -3 };            // yhd_stream::hit.x of a slot in the finish list

YH_DEV int lane_rank(unsigned long long m) {  // set bits of m below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// Appends `value` of the lanes with `pred` to a list of this wave (all lanes of the wave call it together).
YH_DEV int list_push(YH_LDS unsigned short* list, int n, bool pred, int value) {
  unsigned long long m = __ballot(pred);
  if (pred) list[n + lane_rank(m)] = (unsigned short)value;
  return n + (int)__popcll(m);
}
// Closest hit of a finished ray into its slot, RAW as the traversal keeps it (dev_lane.h: the shading stage applies lane_hit
// to the batch it shades, once per hit instead of in every step that retires a ray); returns what it hit.
YH_DEV int publish(const yhd_stream& pl, size_t g, const hit_t& hit, bool hit_lines, unsigned int steps) {
  // ONE sector of the slot's line: the hit and, next to it, its distance and the steps of this ray — a scheduling hint of the pixel's work item,
  // added to the pixel's total by the stage that takes the path next (NOT an atomic add here: device-scope atomics execute at the memory side
  // and drop the slot's line from L2 — measured 1.6x on the whole kernel)
  SLOT_HIT(pl, g)  = yhd_int4{hit.object, hit.slot, __float_as_int(hit.u), __float_as_int(hit.v)};
  ((float*)&SLOT_HIT2(pl, g))[0]        = hit.distance;
  ((unsigned int*)&SLOT_HIT2(pl, g))[1] = steps;
  return hit.object < 0 ? K_MISS : (hit_lines ? K_HAIR : K_SURF);
}
YH_DEV int path_flags(const path_t& ps) { return (ps.bounce & 255) | (ps.hit ? 256 : 0) | (ps.in_medium ? 512 : 0); }

template <bool GENERAL, int WAVES, bool PROF>
__global__ __launch_bounds__(YH_ST_BLOCK, WAVES) void k_stream(const yhd_scene sc, const yhd_scene* sc_dev, const yhd_state st, int nsamples,
    const yhd_stream pl) {
  constexpr int WPB = YH_ST_BLOCK / 64;
  extern __shared__ v4f lds_dyn[];
  const int P = pl.slots_per_wave;
  // LDS: [tables: scene level | camera | small lights | env cdf index][per wave: stack window 64 x YH_LSTACK | who tests which segment (dev_lane.h: COOP) | six lists of P slot ids]
  YH_LDS v4f*   lds_tabs  = (YH_LDS v4f*)lds_dyn;  // dev_trace.h: stage_tables
  const int     wave_lds  = YH_ST_WAVE_LDS(P);
  const int     lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (the wave's index, known to be uniform: the addresses of its lists live in scalar registers)
  YH_LDS unsigned char*  wbase   = (YH_LDS unsigned char*)(lds_tabs + YHD_LDS_TABLES_F4(&sc)) + wib * wave_lds;
  YH_LDS unsigned int*   w_stack = (YH_LDS unsigned int*)wbase;
  YH_LDS unsigned long long* w_cmap = (YH_LDS unsigned long long*)(w_stack + 64 * YH_LSTACK);
  YH_LDS unsigned short* l_ray   = (YH_LDS unsigned short*)(w_stack + 64 * YH_LSTACK + 128);
  YH_LDS unsigned short* l_done  = l_ray + P;
  YH_LDS unsigned short* l_hair  = l_done + P;
  YH_LDS unsigned short* l_surf  = l_hair + P;
  YH_LDS unsigned short* l_fin   = l_surf + P;
  YH_LDS unsigned short* l_free  = l_fin + P;
  const size_t wave_id = (size_t)blockIdx.x * WPB + wib;
  const size_t base    = wave_id * (size_t)P;  // this wave's first slot

  trace_ctx tc;
  tc.sc = &sc, tc.sc_dev = sc_dev, tc.lds_stack = nullptr, tc.stats = nullptr;
  YH_LDS float* lds_cam;
  stage_tables(sc, lds_tabs, threadIdx.x, YH_ST_BLOCK, tc, lds_cam);  // shared by the block's waves
  // slots [0, seeded) of this wave's pool hold pixels already — its own share of the work list, put there by k_stream_seed below —
  // and start in the finish list (their first camera rays); the others are free
  // (readfirstlane: the count is the same for the whole wave, and the compiler has to KNOW that — loaded per lane it makes every list count
  // a vector register and the scheduler's branches divergent: 24 spill reloads in the step loop and 0.7 x on curly-hair, profiles/r05/k_stream_wave_shares.txt)
  const int seeded = __builtin_amdgcn_readfirstlane(pl.wave_fill ? pl.wave_fill[(size_t)blockIdx.x * WPB + wib] : 0);
  for (int s = lane; s < P; s += 64) {
    if (s < seeded) l_fin[s] = (unsigned short)s;
    else l_free[s - seeded] = (unsigned short)s;
  }
  __syncthreads();  // the only workgroup barrier: from here on every wave runs on its own

  if (pl.wave_log && lane == 0) pl.wave_log[2 * wave_id] = wall_clock64();  // when this wave's work begins (written now: nothing to keep in a register)
  lane_stack stk;
  stk.lds = w_stack + lane, stk.ovf = pl.stack_ovf + wave_id * (size_t)pl.ovf_entries * 64 + lane, stk.sp = 0, stk.base = 0;
  tc.ls   = &stk;
  // lists (wave-uniform counts) and the ray this lane holds
  int       n_ray = 0, n_done = 0, n_hair = 0, n_surf = 0, n_fin = seeded, n_free = P - seeded;
  int       my_group = (int)(blockIdx.x % (unsigned)st.num_groups), groups_done = 0;  // item group being taken from (yh_device.h)
  bool      nomore = false;
  bool      have = false;
  int       slot = 0;
  lane_trav t;
  lane_begin(sc, t, mk3(0.0f), mk3(1.0f), -1);

  enum { A_ITEMS, A_SORT, A_FINISH, A_HAIR, A_SURF, A_TRACE };
  // One stage per trip, each stage's code exactly once in the kernel.
  while (true) {
    const int nact = (int)__popcll(__ballot(have));
    // Hair shading is the expensive stage (thousands of instructions): full batches of 64, a partial one only when
    // the rays at hand (plus the samples the finish stage is about to start) cannot fill the lanes. Surface shading
    // and finishing are a few hundred instructions: whatever has gathered since the last trace stage, so that no
    // path waits long for company. Finishing comes after shading (which ends paths) and runs once per round.
    int act;
    const bool flush = n_ray + nact < 64;
    if (!nomore && n_free >= 64) act = A_ITEMS;
    else if (n_done > 0) act = A_SORT;
    else if (n_fin >= 64 || (flush && n_fin > 0)) act = A_FINISH;
    else if (n_hair >= 64 || (flush && n_hair > 0)) act = A_HAIR;
    else if (n_surf >= 64 || (flush && n_surf > 0)) act = A_SURF;
    else if (n_ray > 0 || nact > 0) act = A_TRACE;
    else break;  // every slot is free and the work items are used up
    unsigned long long pc0 = 0, p_steps = 0, p_busy = 0, p_batch = 0;
    if (PROF) pc0 = clock64();

    if (act == A_ITEMS) {
      // ---- items: free slots take the pixels of the next work items (4 items = 64 pixels) ------------------
      // the next items of this workgroup's group (its XCD's image region); a group that is used up hands over to the next
      int       t0 = 0, got = 0;
      const int want = 4;
      while (true) {
        int c = 0;
        if (lane == 0) c = atomicAdd(st.tile_cursor + 16 * my_group, want);
        c   = __builtin_amdgcn_readfirstlane(c);
        t0  = st.group_begin[my_group] + c;
        got = max(0, min(want, st.group_begin[my_group + 1] - t0));
        if (got > 0 || ++groups_done >= st.num_groups) break;
        my_group = (my_group + 1) % st.num_groups;
      }
      if (got == 0) nomore = true;
      int pixel = -1, item = 0;
      if (lane < 16 * got) {
        item     = st.tiles[t0 + (lane >> 4)];
        int tile = item >> 2, part = item & 3, pq = lane & 15;
        int i    = (tile % st.tiles_x) * YH_TILE + (part & 1) * 4 + (pq & 3);
        int j    = (tile / st.tiles_x) * YH_TILE + (part >> 1) * 4 + (pq >> 2);
        pixel    = (i < st.width && j < st.height) ? j * st.width + i : -1;
      }
      const bool               valid = pixel >= 0;
      const unsigned long long m     = __ballot(valid);
      int                      sl    = 0;
      if (valid) {
        sl                = l_free[n_free - 1 - lane_rank(m)];
        const size_t   g  = base + sl;
        const uint64_t rs = st.rng_state[pixel], ri = st.rng_inc[pixel];
        SLOT_RADIANCE(pl, g)    = yhd_float4{0.0f, 0.0f, 0.0f, __int_as_float(pixel)};
        SLOT_OWN(pl, g)         = yhd_int4{nsamples, item, (int)(unsigned)ri, (int)(unsigned)(ri >> 32)};
        SLOT_RNGW(pl, g)        = yhd_int4{(int)(unsigned)rs, (int)(unsigned)(rs >> 32), 0, 0};
        SLOT_HIT(pl, g)         = yhd_int4{H_NEW, 0, 0, 0};
      }
      n_free -= (int)__popcll(m);
      n_fin = list_push(l_fin, n_fin, valid, sl);
    } else if (act == A_SORT) {
      // ---- sort: the rays the trace stage finished, by what they hit ----------------------------------------
      for (int i0 = 0; i0 < n_done; i0 += 64) {
        const bool on = i0 + lane < n_done;
        const int  e  = on ? (int)l_done[i0 + lane] : 0;
        const int  sl = e & 0xFFF;
        int        kd = on ? (e >> 12) : -1;
        if (__ballot(kd == K_REDO) != 0) {
          if (kd == K_REDO) {  // axis-parallel ray: the reference's compare-and-select box test (dev_trace.h)
            const size_t g = base + sl;
            yhd_float4 o = SLOT_RAY_O(pl, g), d = SLOT_RAY_D(pl, g);
            lane_exact_result r = lane_trace_exact(sc_dev, tc.lds_scene, stk.lds, stk.ovf, stk.sp, stk.base, f3{o.x, o.y, o.z},
                f3{d.x, d.y, d.z}, -1);
            stk.base = r.base;
            kd       = publish(pl, base + sl, r.hit, r.hit_lines != 0, 1u);
          }
        }
        n_hair = list_push(l_hair, n_hair, kd == K_HAIR, sl);
        n_surf = list_push(l_surf, n_surf, kd == K_SURF, sl);
        n_fin  = list_push(l_fin, n_fin, kd == K_MISS, sl);
      }
      n_done = 0;
    } else if (act == A_FINISH) {
      // ---- finish: account the sample that ended, start the pixel's next one ---------------------------------
      const int  cnt = min(n_fin, 64);
      const bool on  = lane < cnt;
      const int  sl  = on ? l_fin[n_fin - cnt + lane] : 0;
      n_fin -= cnt;
      if (PROF) p_batch = (unsigned long long)cnt;
      bool next = false, freed = false;
      if (on) {
        const size_t     g   = base + sl;
        const yhd_float4 rad = SLOT_RADIANCE(pl, g);
        const yhd_int4   own = SLOT_OWN(pl, g), rw = SLOT_RNGW(pl, g);
        const int        h = SLOT_HIT(pl, g).x, p = __float_as_int(rad.w), left = own.x;
        unsigned int     work = (unsigned int)rw.z;
        if (h == H_MISS) work += (unsigned int)SLOT_HIT2(pl, g).y;  // the ray that missed (see publish)
        if (h != H_NEW) {  // trace_sample's tail (pt.cpp:1683-1688)
          const yhd_float4 w = SLOT_WEIGHT(pl, g);
          path_t           ps;
          ps.radiance = f3{rad.x, rad.y, rad.z};
          ps.hit      = (__float_as_int(w.w) & 256) != 0;
          if (h == H_MISS) {  // pt.cpp:1397-1400
            const yhd_float4 d = SLOT_RAY_D(pl, g);
            ps.radiance        = ps.radiance + f3{w.x, w.y, w.z} * eval_environment<false>(tc, f3{d.x, d.y, d.z});
          }
          yhd_float4 acc = st.accum[p];
          path_end(ps, st.clamp, acc);
          st.accum[p] = acc;
        }
        rng_t rng;
        rng.state = (uint64_t)(unsigned)rw.x | ((uint64_t)(unsigned)rw.y << 32);
        rng.inc   = (uint64_t)(unsigned)own.z | ((uint64_t)(unsigned)own.w << 32);
        if (left > 0) {  // the pixel's next sample (trace_sample, pt.cpp:1676-1682)
          float lu = rand1f(rng), lv = rand1f(rng);
          float pu = rand1f(rng), pv = rand1f(rng);
          yhd_camera cam;
          for (int k = 0; k < 12; k++) cam.frame[k] = lds_cam[k];
          cam.lens = lds_cam[12], cam.film_x = lds_cam[13], cam.film_y = lds_cam[14], cam.focus = lds_cam[15], cam.aperture = lds_cam[16];
          ray_t r = sample_camera_lane(cam, p % st.width, p / st.width, st.width, st.height, pu, pv, lu, lv);
          SLOT_RAY_O(pl, g)    = yhd_float4{r.o.x, r.o.y, r.o.z, 0.0f};
          SLOT_RAY_D(pl, g)    = yhd_float4{r.d.x, r.d.y, r.d.z, 0.0f};
          SLOT_WEIGHT(pl, g)   = yhd_float4{1.0f, 1.0f, 1.0f, __int_as_float(0)};
          SLOT_RNGW(pl, g)     = yhd_int4{(int)(unsigned)rng.state, (int)(unsigned)(rng.state >> 32), (int)work, 0};
          SLOT_RADIANCE(pl, g) = yhd_float4{0.0f, 0.0f, 0.0f, rad.w};
          SLOT_OWN(pl, g)      = yhd_int4{left - 1, own.y, own.z, own.w};
          next           = true;
        } else {  // the pixel has all its samples: hand its stream back, report its work, free the slot
          st.rng_state[p] = rng.state;
          if (work) atomicAdd(&st.tile_cost[own.y], work);
          freed = true;
        }
      }
      n_ray  = list_push(l_ray, n_ray, next, sl);
      n_free = list_push(l_free, n_free, freed, sl);
    } else if (act == A_HAIR || act == A_SURF) {
      // ---- shade: up to 64 hits of ONE kind, one path per lane (trace_path's loop body, pt.cpp:1395-1508) -------
      YH_LDS unsigned short* list = act == A_HAIR ? l_hair : l_surf;
      const int  n   = act == A_HAIR ? n_hair : n_surf;
      const int  cnt = min(n, 64);
      const bool on  = lane < cnt;
      const int  sl  = on ? list[n - cnt + lane] : 0;
      if (act == A_HAIR) n_hair -= cnt;
      else n_surf -= cnt;
      if (PROF) p_batch = (unsigned long long)cnt;
      bool alive = false;
      if (on) {
        const size_t g = base + sl;
        const yhd_float4 o = SLOT_RAY_O(pl, g), d = SLOT_RAY_D(pl, g), w = SLOT_WEIGHT(pl, g), rad = SLOT_RADIANCE(pl, g);
        const yhd_int4   h = SLOT_HIT(pl, g), h2 = SLOT_HIT2(pl, g), rw = SLOT_RNGW(pl, g), own = SLOT_OWN(pl, g);
        path_t     ps;
        ps.ray      = ray_t{f3{o.x, o.y, o.z}, f3{d.x, d.y, d.z}, ray_eps, flt_max};
        ps.weight   = f3{w.x, w.y, w.z}, ps.radiance = f3{rad.x, rad.y, rad.z};
        const int fl = __float_as_int(w.w);
        ps.bounce = fl & 255, ps.hit = (fl & 256) != 0, ps.in_medium = (fl & 512) != 0;
        ps.medium_mem = GENERAL ? pl.medium + 2 * g : nullptr;  // (read and written where path_step needs it: dev_path.h)
        hit_t isec;
        isec.object = h.x, isec.slot = h.y, isec.u = __int_as_float(h.z), isec.v = __int_as_float(h.w), isec.distance = __int_as_float(h2.x);
        isec = lane_hit_retest(tc, isec, act == A_HAIR, ps.ray.o, ps.ray.d);  // (hair batches hold the hits on lines: their uv from the test itself, dev_lane.h)
        rng_t rng;
        rng.state = (uint64_t)(unsigned)rw.x | ((uint64_t)(unsigned)rw.y << 32);
        rng.inc   = (uint64_t)(unsigned)own.z | ((uint64_t)(unsigned)own.w << 32);
        alive     = path_step<false, 64, GENERAL>(tc, ps, isec, rng, st.bounces);
        // sector 1, every bounce: weight + flags | the stream's state + the steps of the ray just shaded
        SLOT_WEIGHT(pl, g) = yhd_float4{ps.weight.x, ps.weight.y, ps.weight.z, __int_as_float(path_flags(ps))};
        SLOT_RNGW(pl, g)   = yhd_int4{(int)(unsigned)rng.state, (int)(unsigned)(rng.state >> 32), rw.z + h2.y, 0};
        // sector 3 only when the bounce met an emitter (an area light, or a textured emission): the radiance is otherwise what it was
        if (ps.radiance.x != rad.x || ps.radiance.y != rad.y || ps.radiance.z != rad.z) SLOT_RADIANCE(pl, g) = yhd_float4{ps.radiance.x, ps.radiance.y, ps.radiance.z, rad.w};
        if (alive) {  // sector 0: the next ray
          SLOT_RAY_O(pl, g) = yhd_float4{ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, 0.0f};
          SLOT_RAY_D(pl, g) = yhd_float4{ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, 0.0f};
        } else {
          SLOT_HIT(pl, g).x = H_ENDED;
        }
      }
      n_ray = list_push(l_ray, n_ray, on && alive, sl);
      n_fin = list_push(l_fin, n_fin, on && !alive, sl);
    } else {
      // ---- trace: one ray per lane; finished lanes refill from the ray list; leaves with rays suspended ---------
      unsigned long long tacc[5] = {};  // PROF: shader-clock cycles of the five parts of a step (dev_lane.h), this wave's sums
      float pc[2 * LP_COUNT] = {};  // PROF: per branch of lane_step, this lane's share of the wave steps that ran it and of the lanes in them
      while (true) {
        YH_MARK("trace_refill");
        {  // refill: the (divergent) refill code runs only when enough lanes are idle, or none is busy
          const unsigned long long idle  = __ballot(!have);
          const int                nidle = (int)__popcll(idle);
          if (n_ray > 0 && (nidle >= YH_REFILL_LANES || nidle == 64)) {
            if (!have && lane_rank(idle) < n_ray) {
              slot           = l_ray[n_ray - 1 - lane_rank(idle)];
              const size_t g = base + slot;
              yhd_float4 o = SLOT_RAY_O(pl, g), d = SLOT_RAY_D(pl, g);
              lane_begin(sc, t, f3{o.x, o.y, o.z}, f3{d.x, d.y, d.z}, -1);
              have = true;
            }
            n_ray -= min(nidle, n_ray);
          }
        }
        const int busy = (int)__popcll(__ballot(have));
        if (busy == 0) break;
        // Leave for the shading stages when the ray list is dry and either a full batch of hair hits has gathered
        // or few lanes are busy; the unfinished rays stay in their lanes.
        if (n_ray == 0 && busy <= pl.suspend_lanes && (n_done | n_hair | n_surf | n_fin) != 0) break;
        if (PROF) p_steps++, p_busy += (unsigned long long)busy;
        bool fin  = false;
        int  kind = 0;
        {
          bool redo = false;
          if (lane_step<false, PROF, true>(tc, t, stk, 0, redo, PROF && pl.prof_parts_only ? nullptr : pc, have, w_cmap, PROF ? tacc : nullptr)) {  // (every lane: the wave tests its line leaves together)
            YH_MARK("trace_retire");
            have = false, fin = true;
            if (redo) {
              kind   = K_REDO;  // traced again by the exact form in the sort stage
              stk.sp = 0, stk.base = 0;
            } else {
              kind = publish(pl, base + slot, t.hit, t.hit_lines, t.steps);
            }
          }
        }
        YH_MARK("trace_lists");
        if (__ballot(fin) != 0) n_done = list_push(l_done, n_done, fin, slot | (kind << 12));
        YH_MARK("trace_loop_end");
      }
      if (PROF) {
        if (lane == 0)
          for (int k = 0; k < 5; k++) atomicAdd(&pl.prof[52 + k], tacc[k]);
        for (int k = 0; k < 2 * LP_COUNT; k++) {
          float v = pc[k];
          for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
          if (lane == 0 && v > 0) atomicAdd(&pl.prof[32 + k], (unsigned long long)(v + 0.5f));
        }
      }
    }
    if (PROF && lane == 0) {  // developer build: where the wave's time goes (tools/stream_prof.py)
      atomicAdd(&pl.prof[act], (unsigned long long)(clock64() - pc0));
      atomicAdd(&pl.prof[8 + act], 1ull);
      atomicAdd(&pl.prof[16 + act], p_batch);
      if (act == A_TRACE) atomicAdd(&pl.prof[24], p_steps), atomicAdd(&pl.prof[25], p_busy);
    }
  }
  if (pl.wave_log && lane == 0) pl.wave_log[2 * wave_id + 1] = wall_clock64();  // ... and ends: what the next hand-out sizes the waves' shares by
}


// A WAVE'S OWN SHARE of the work list (yh_device.h: yhd_stream::wave_begin; one-generation images, host/launch_plan.cpp:
// deal_shares_by_speed): one wavefront here per wavefront of k_stream puts the pixels of the share's entries into the first slots of that
// wave's pool — what k_stream's items stage does with the entries it takes from the cursor — and says how many (wave_fill). A kernel of
// its own, launched in front of k_stream: k_stream itself only learns how many of its slots are filled.
__global__ __launch_bounds__(64) void k_stream_seed(const yhd_state st, int nsamples, const yhd_stream pl, int waves) {
  const int wave_id = blockIdx.x, lane = threadIdx.x;
  if (wave_id >= waves) return;
  const size_t base = (size_t)wave_id * (size_t)pl.slots_per_wave;
  int          n    = 0;
  for (int t0 = pl.wave_begin[wave_id], t1 = pl.wave_begin[wave_id + 1]; t0 < t1; t0 += 4) {  // four entries = 64 pixels (padded with -1 = no item)
    const int item  = st.tiles[t0 + (lane >> 4)];
    const int tile  = item >> 2, part = item & 3, pq = lane & 15;
    const int i     = (tile % st.tiles_x) * YH_TILE + (part & 1) * 4 + (pq & 3);
    const int j     = (tile / st.tiles_x) * YH_TILE + (part >> 1) * 4 + (pq >> 2);
    const int pixel = (item >= 0 && i < st.width && j < st.height) ? j * st.width + i : -1;
    const bool               valid = pixel >= 0;
    const unsigned long long m     = __ballot(valid);
    if (valid && n + lane_rank(m) < pl.slots_per_wave) {  // (the host sizes the pool for the largest share)
      const size_t   g  = base + (size_t)(n + lane_rank(m));
      const uint64_t rs = st.rng_state[pixel], ri = st.rng_inc[pixel];
      SLOT_RADIANCE(pl, g) = yhd_float4{0.0f, 0.0f, 0.0f, __int_as_float(pixel)};
      SLOT_OWN(pl, g)      = yhd_int4{nsamples, item, (int)(unsigned)ri, (int)(unsigned)(ri >> 32)};
      SLOT_RNGW(pl, g)     = yhd_int4{(int)(unsigned)rs, (int)(unsigned)(rs >> 32), 0, 0};
      SLOT_HIT(pl, g)      = yhd_int4{H_NEW, 0, 0, 0};
    }
    n += (int)__popcll(m);
  }
  if (lane == 0) pl.wave_fill[wave_id] = min(n, pl.slots_per_wave);
}

// ---------------------------------------------------------------------------------------------------------------
// Closest-hit batches with one lane per ray (yh_intersect_batch on large batches): persistent wavefronts, the trace
// stage of k_stream on its own — a lane that finishes its ray takes the next one of the batch (refill when enough
// lanes are idle) — and nothing of the shading code around it, so the kernel fits 64-80 registers and runs at 6-8
// waves per SIMD. Same lane_step, same closest hits as the quad form (k_intersect). tmin of every ray = ray_eps.
// ---------------------------------------------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_intersect_lanes(const yhd_scene sc, const yhd_scene* sc_dev, int n, const float* rays,
    int* cursor, unsigned int* stack_ovf, int ovf_entries, int* object, int* element, float* uv, float* dist) {
  extern __shared__ v4f lds_dyn[];
  YH_LDS v4f* lds_tabs = (YH_LDS v4f*)lds_dyn;
  const int   lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  YH_LDS unsigned int* w_stack = (YH_LDS unsigned int*)(lds_tabs + YHD_LDS_TABLES_F4(&sc)) + wib * (64 * YH_LSTACK + 128);
  YH_LDS unsigned long long* w_cmap = (YH_LDS unsigned long long*)(size_t)__builtin_amdgcn_readfirstlane((unsigned int)(size_t)(w_stack + 64 * YH_LSTACK));
  trace_ctx tc;
  tc.sc = &sc, tc.sc_dev = sc_dev, tc.lds_stack = nullptr, tc.stats = nullptr;
  YH_LDS float* lds_cam;
  stage_tables(sc, lds_tabs, threadIdx.x, 256, tc, lds_cam);
  __syncthreads();
  const size_t wave_id = (size_t)blockIdx.x * 4 + wib;
  lane_stack   stk;
  stk.lds = w_stack + lane, stk.ovf = stack_ovf + wave_id * (size_t)ovf_entries * 64 + lane, stk.sp = 0, stk.base = 0;
  tc.ls = &stk;
  lane_trav t;
  lane_begin(sc, t, mk3(0.0f), mk3(1.0f), -1);
  bool have = false, dry = false;
  int  ray = 0;
  while (true) {
    const unsigned long long idle  = __ballot(!have);
    const int                nidle = (int)__popcll(idle);
    if (!dry && (nidle >= YH_REFILL_LANES || nidle == 64)) {
      int first = 0;
      if (lane == 0) first = atomicAdd(cursor, nidle);
      first = __builtin_amdgcn_readfirstlane(first);
      if (first >= n) dry = true;
      const int mine = first + lane_rank(idle);
      if (!have && mine < n) {
        ray            = mine;
        const float* r = rays + 8 * (size_t)ray;
        lane_begin(sc, t, ld3(r), ld3(r + 3), -1);
        t.tmax = r[7];
        have   = true;
      }
    }
    if (__ballot(have) == 0) {
      if (dry) break;
      continue;
    }
    {
      bool redo = false;
      if (lane_step<false, false, true>(tc, t, stk, 0, redo, nullptr, have, w_cmap)) {
        have = false;
        hit_t h = lane_hit_retest(tc, t.hit, t.hit_lines, t.ro, t.rd);
        if (redo) {  // axis-parallel ray: the reference's compare-and-select box test throughout
          stk.sp = 0, stk.base = 0;
          const float*      r = rays + 8 * (size_t)ray;
          lane_exact_result e = lane_trace_exact(sc_dev, tc.lds_scene, stk.lds, stk.ovf, 0, 0, ld3(r), ld3(r + 3), -1);
          stk.base = e.base;
          h        = lane_hit_retest(tc, e.hit, e.hit_lines != 0, ld3(r), ld3(r + 3));
          if (h.object >= 0 && h.distance > r[7]) h.object = -1, h.slot = -1, h.u = 0, h.v = 0, h.distance = 0;  // (the exact form starts from tmax = flt_max)
        }
        object[ray] = h.object, element[ray] = hit_element(sc, h);
        uv[2 * ray] = h.u, uv[2 * ray + 1] = h.v, dist[ray] = h.distance;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The test records of the traversal kernels' array (yh_device.h: yhd_scene::lane_blob): the ray-test halves of the leaf-ordered
// primitive records, one thread per record, HBM-streaming. (The nodes of that array are written by csrc/bvh_gpu.hip: k_wide_collapse.)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_lane_tests(const yhd_float4* __restrict__ prims, yhd_float4* __restrict__ out, int kind, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (kind == YH_KIND_LINES) {
    out[2 * (size_t)i] = prims[4 * (size_t)i], out[2 * (size_t)i + 1] = prims[4 * (size_t)i + 1];
  } else {
    out[4 * (size_t)i] = prims[6 * (size_t)i], out[4 * (size_t)i + 1] = prims[6 * (size_t)i + 1], out[4 * (size_t)i + 2] = prims[6 * (size_t)i + 2];
    out[4 * (size_t)i + 3] = yhd_float4{0, 0, 0, 0};
  }
}
extern "C" {

int yhk_lane_tests(const yhd_float4* prims, yhd_float4* blob, int kind, int prim_base, int num_prims, long long test_off, hipStream_t stream) {
  if (num_prims > 0)
    hipLaunchKernelGGL(k_lane_tests, dim3((num_prims + 255) / 256), dim3(256), 0, stream, prims + prim_base, blob + 2 * test_off, kind, num_prims);
  return (int)hipGetLastError();
}

// waves: 4, 6 or 8 per SIMD (the register budget: 128 / 80 / 64)
typedef void (*lanes_kernel_t)(const yhd_scene, const yhd_scene*, int, const float*, int*, unsigned int*, int, int*, int*, float*, float*);
static lanes_kernel_t lanes_kernel(int waves) { return waves >= 8 ? k_intersect_lanes<8> : waves >= 6 ? k_intersect_lanes<6> : k_intersect_lanes<4>; }
int yhk_intersect_lanes_lds(const yhd_scene* sc) { return YHD_LDS_TABLES_F4(sc) * 16 + 4 * (64 * YH_LSTACK * 4 + 64 * 8); }
int yhk_intersect_lanes_occupancy(const yhd_scene* sc, int waves) {
  int blocks = 0, lds = yhk_intersect_lanes_lds(sc);
  lanes_kernel_t k = lanes_kernel(waves);
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k, 256, lds) != hipSuccess) return 0;
  return blocks;
}
int yhk_intersect_lanes(const yhd_scene* sc, const yhd_scene* sc_dev, int n, const float* rays, int* cursor, unsigned int* stack_ovf,
    int ovf_entries, int* object, int* element, float* uv, float* dist, int waves, int grid_blocks, hipStream_t stream) {
  int            lds = yhk_intersect_lanes_lds(sc);
  lanes_kernel_t k   = lanes_kernel(waves);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid_blocks), dim3(256), lds, stream, *sc, sc_dev, n, rays, cursor, stack_ovf, ovf_entries, object, element, uv, dist);
  return (int)hipGetLastError();
}

typedef void (*stream_kernel_t)(const yhd_scene, const yhd_scene*, const yhd_state, int, const yhd_stream);
static stream_kernel_t stream_kernel(bool general, bool prof = false) {
  if (prof && !general) return k_stream<false, YH_ST_WAVES, true>;
  return general ? k_stream<true, YH_ST_WAVES_GENERAL, false> : k_stream<false, YH_ST_WAVES, false>;
}
int yhk_stream_block_threads(void) { return YH_ST_BLOCK; }
int yhk_stream_lds_bytes(int tables_f4, int slots_per_wave) {
  return tables_f4 * 16 + (YH_ST_BLOCK / 64) * YH_ST_WAVE_LDS(slots_per_wave);
}
int yhk_stream_occupancy(int lds_bytes, int general) {
  int             blocks = 0;
  stream_kernel_t kern   = stream_kernel(general != 0);
  if (lds_bytes > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kern, YH_ST_BLOCK, lds_bytes) != hipSuccess) return 0;
  return blocks;
}
int yhk_stream(const yhd_scene* sc, const yhd_scene* sc_dev, const yhd_state* st, int nsamples, const yhd_stream* pl, int grid_blocks,
    hipStream_t stream) {
  if (pl->suspend_lanes < 0 || pl->suspend_lanes >= 64) return (int)hipErrorInvalidValue;
  if (pl->wave_begin && pl->wave_fill) {  // the waves' own shares into their pools first
    const int waves = grid_blocks * (YH_ST_BLOCK / 64);
    hipLaunchKernelGGL(k_stream_seed, dim3(waves), dim3(64), 0, stream, *st, nsamples, *pl, waves);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  int             lds  = yhk_stream_lds_bytes(YHD_LDS_TABLES_F4(sc), pl->slots_per_wave);
  stream_kernel_t kern = stream_kernel(sc->general_materials != 0, pl->prof != nullptr);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(grid_blocks), dim3(YH_ST_BLOCK), lds, stream, *sc, sc_dev, *st, nsamples, *pl);
  return (int)hipGetLastError();
}
}
