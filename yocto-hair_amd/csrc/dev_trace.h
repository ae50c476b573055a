// dev_trace.h — PCG32, ray / primitive tests and the two-level BVH traversal.
//
// Restates math.h:1396-1442 (PCG32), math.h:3426-3505,3544-3554
// (intersect_line / _triangle / _bbox) and pt.cpp:821-1053
// (intersect_shape_bvh / intersect_scene_bvh / intersect_instance_bvh).
//
// MI355X-first differences from the reference's nested, pointer-chasing
// loops (results are unchanged):
//   * QUADS: four adjacent lanes own ONE ray. A hair path is a long serial
//     chain (its pixel's PCG32 stream is sequential) and a lone wavefront
//     issues at most one vector instruction every ~4 cycles, so the cost of a
//     ray is the number of instructions on its chain. With a quad per ray a
//     node step is one box test per lane (not four per lane) and a leaf step
//     is one primitive test per lane (not up to four), exchanged with DPP
//     quad permutes — no LDS, no memory.
//   * the shape BVH is 4-wide: two levels of the reference's binary tree per
//     128-byte node (host/bvh_build.h), one dependent fetch per step; children
//     are visited in the order the binary traversal would visit them.
//   * ONE flattened loop with one stack for both levels: entries are tagged
//     {wide node, leaf, scene node, enter-instance}; the scene level's leaf
//     loop `for idx in start..start+num` becomes enter-instance entries pushed
//     in reverse order, so instances are visited in the reference's order and
//     `tmax` shrinks identically.
//   * the stack lives in LDS, one column per quad; inverse object frames are
//     precomputed at upload (pt.cpp:1012-1013 recomputes a 3x3 adjugate
//     inverse per ray per object); leaf primitives are 16-byte-aligned records
//     in leaf order (yh_device.h).
#ifndef YH_DEV_TRACE_H_
#define YH_DEV_TRACE_H_
#include "dev_math.h"

namespace yhd {

// ---------------------------------------------------------------------------
// PCG32 (math.h:1396-1442)
// ---------------------------------------------------------------------------
struct rng_t {
  uint64_t state, inc;
};
YH_DEV uint32_t advance_rng(rng_t& rng) {
  uint64_t old        = rng.state;
  rng.state           = old * 6364136223846793005ULL + rng.inc;
  uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot        = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
YH_DEV float rand1f(rng_t& rng) {
  return __uint_as_float((advance_rng(rng) >> 9) | 0x3f800000u) - 1.0f;
}
YH_DEV rng_t make_rng(uint64_t seed, uint64_t seq) {
  rng_t rng;
  rng.state = 0U;
  rng.inc   = (seq << 1u) | 1u;
  advance_rng(rng);
  rng.state += seed;
  advance_rng(rng);
  return rng;
}

// ---------------------------------------------------------------------------
// Rays and primitive tests
// ---------------------------------------------------------------------------
struct ray_t {
  f3    o, d;
  float tmin, tmax;
};
YH_DEV ray_t mkray(f3 o, f3 d) { return ray_t{o, d, ray_eps, flt_max}; }

struct hit_t {
  int   object, slot;  // object -1 on miss; slot = leaf-order index of the primitive in its shape
  float u, v, distance;
};

// math.h:3426-3469
// STRAIGHT: the reference's three early returns as one predicate. The lanes of a wave test
// different primitives, so in dense hair an early return almost never skips the rest for the whole
// wave while each one costs an exec-mask save / branch / restore on the scalar unit (dense launch
// shape: +2 % on C2); in C1 leaf steps run with a few lanes and the returns do skip (-3 %), so
// the 512-thread shape keeps them. A zero det makes t and s inf or NaN and every comparison false,
// as the first return would.
// RAW (the traversal loop below): `a` = dot(rd, rd) is handed in (the same for every segment a ray meets inside one object) and an
// accepted hit returns vv = d2 and rr = r instead of vv = sqrt(d2) / r — the loop evaluates that once, for the hit that survives
// (same operands, same operations: same bits).
template <bool STRAIGHT = false, bool RAW = false>
YH_DEV bool intersect_line(f3 ro, f3 rd, float tmin, float tmax, f3 p0, f3 p1, float r0,
    float r1, float& uu, float& vv, float& dist, float a_in = 0.0f, float* rr = nullptr) {
  f3    u = rd, v = p1 - p0, w = ro - p0;
  float a = RAW ? a_in : dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w);
  float det = a * c - b * b;
  if (STRAIGHT) {
    float t   = (b * e - c * d) / det;
    float s   = (a * e - b * d) / det;
    bool  ok  = det != 0 && !(t < tmin || t > tmax);
    s         = fclamp(s, 0.0f, 1.0f);
    f3    pr  = ro + rd * t;
    f3    pl  = p0 + (p1 - p0) * s;
    f3    prl = pr - pl;
    float d2  = dot(prl, prl);
    float r   = r0 * (1 - s) + r1 * s;
    ok        = ok && !(d2 > r * r);
    if (RAW) {
      uu = s, vv = d2, *rr = r, dist = t;
      return ok;
    }
    if (ok) uu = s, vv = sqrtf(d2) / r, dist = t;
    return ok;
  }
  if (det == 0) return false;
  float t = (b * e - c * d) / det;
  float s = (a * e - b * d) / det;
  if (t < tmin || t > tmax) return false;
  s        = fclamp(s, 0.0f, 1.0f);
  f3    pr = ro + rd * t;
  f3    pl = p0 + (p1 - p0) * s;
  f3    prl = pr - pl;
  float d2  = dot(prl, prl);
  float r   = r0 * (1 - s) + r1 * s;
  if (d2 > r * r) return false;
  if (RAW) uu = s, vv = d2, *rr = r;
  else uu = s, vv = sqrtf(d2) / r;
  dist = t;
  return true;
}
// math.h:3472-3505
YH_DEV bool intersect_triangle(f3 ro, f3 rd, float tmin, float tmax, f3 p0, f3 p1, f3 p2,
    float& uu, float& vv, float& dist) {
  f3    edge1 = p1 - p0, edge2 = p2 - p0;
  f3    pvec = cross(rd, edge2);
  float det  = dot(edge1, pvec);
  if (det == 0) return false;
  float inv_det = 1.0f / det;
  f3    tvec    = ro - p0;
  float u       = dot(tvec, pvec) * inv_det;
  if (u < 0 || u > 1) return false;
  f3    qvec = cross(tvec, edge1);
  float v    = dot(rd, qvec) * inv_det;
  if (v < 0 || u + v > 1) return false;
  float t = dot(edge2, qvec) * inv_det;
  if (t < tmin || t > tmax) return false;
  uu = u, vv = v;
  dist = t;
  return true;
}
// math.h:3544-3554
YH_DEV bool intersect_bbox(f3 ro, f3 dinv, float tmin_, float tmax_, f3 bmin, f3 bmax) {
  f3    it_min = (bmin - ro) * dinv;
  f3    it_max = (bmax - ro) * dinv;
  f3    tmin   = {fmin_(it_min.x, it_max.x), fmin_(it_min.y, it_max.y), fmin_(it_min.z, it_max.z)};
  f3    tmax   = {fmax_(it_min.x, it_max.x), fmax_(it_min.y, it_max.y), fmax_(it_min.z, it_max.z)};
  float t0     = fmax_(hmax(tmin), tmin_);
  float t1     = fmin_(hmin(tmax), tmax_);
  t1 *= 1.00000024f;
  return t0 <= t1;
}

// The same test when no operand can be a NaN (finite ray origin and finite 1 / d, i.e. no zero
// direction component: 0 * inf is the only way the slabs produce one). Then `(a < b) ? a : b` and
// the hardware's min / max pick the same value up to the sign of a zero, which no later
// comparison distinguishes, and the test is 3 + 3 min / max and two 3-input min / max instead of a
// dozen compare + select pairs. Callers choose it with a WAVE-UNIFORM condition (a per-lane select
// makes the compiler evaluate both forms).
YH_DEV bool intersect_bbox_nonan(f3 ro, f3 dinv, float tmin_, float tmax_, f3 bmin, f3 bmax) {
  f3    it_min = (bmin - ro) * dinv;
  f3    it_max = (bmax - ro) * dinv;
  float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(it_min.x, it_max.x), __builtin_fminf(it_min.y, it_max.y)),
      __builtin_fmaxf(__builtin_fminf(it_min.z, it_max.z), tmin_));
  float t1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(it_min.x, it_max.x), __builtin_fmaxf(it_min.y, it_max.y)),
      __builtin_fminf(__builtin_fmaxf(it_min.z, it_max.z), tmax_));
  t1 *= 1.00000024f;
  return t0 <= t1;
}

// ---------------------------------------------------------------------------
// Traversal
// ---------------------------------------------------------------------------
#define YH_TAG_SHAPE 0u          /* entry = global wide-node index            */
#define YH_TAG_SCENE 0x40000000u /* entry = scene-node index                 */
#define YH_TAG_ENTER 0x80000000u /* entry = object id to enter               */
#define YH_TAG_LEAF 0xC0000000u  /* entry = count << 27 | first leaf slot    */
#define YH_TAG_MASK 0xC0000000u
#define YH_NONE 0xFFFFFFFFu

// Traversal stack: at most YH_QSTACK entries per quad in LDS (the kernels reserve what the scene's trees need,
// yhd_scene::stack_entries), column `quad` of a [depth][quads] array (conflict-free; the four lanes of a quad
// read / write the same word). yh_upload_scene refuses scenes whose trees could need more.
#define YH_QSTACK 96
// THE HIT RECORD (round 5). A leaf step reduces its lanes' hits to the survivor with DPP exchanges; until round 4 the survivor's u, v
// travelled along (four more exchanges and four more selects per merge stage) and v = sqrt(d2) / r of a line was evaluated by every
// accepted test. Now only the KEY (t, place in the leaf) is merged, the lane that holds the survivor writes its u, raw v (d2 on a
// line) and r into the first YH_HITROWS rows of the path's stack column, and the loop reads them back once, when the ray ends:
// ~24 vector instructions less in every leaf step with a hit (line leaves run in 34 % of C1's and 75 % of hair-curls' trips).
// The traversal stack proper starts YH_HITROWS rows into the column; every launcher reserves entries + YH_HITROWS rows.
// Measured, interleaved on one box (profiles/r05/hit_record_ab.txt): hair-curls (k_trace<256 x 5>) 443 -> 462 Msamples/s (+4.2 %), straight-hair with
// quads +3 %, C1 side by side 13.4 -> 13.0 ms per 64 spp (+2 %); bit-identical.
#define YH_HITROWS 4
#define YH_LDS __attribute__((address_space(3)))
typedef float v4f __attribute__((ext_vector_type(4)));

struct lane_stack;  // dev_lane.h
struct trace_ctx {
  const yhd_scene*      sc;
  YH_LDS unsigned int*  lds_stack;  // this quad's LDS stack column
  // LDS copy of the scene-level tables (k_trace stages them per block when they fit;
  // NULL: read sc.objects / sc.scene_nodes / sc.scene_prims from memory):
  // [objects: 8 float4 each][scene BVH nodes: 2 float4 each][scene BVH primitives]
  const YH_LDS v4f*     lds_scene;
  struct stats_t*       stats;      // per-lane work counters of the instrumented build, else NULL
  const YH_LDS v4f*     lds_lights; // LDS copy of sc.light_table (small area lights), or nullptr
  const YH_LDS float*   lds_envtab; // LDS copy of sc.env_tab (coarse index of an environment's texel cdf), or nullptr
  const YH_LDS v4f*     lds_mats;   // LDS copy of sc.materials (YH_MATERIAL_F4 float4 each), or nullptr
  lane_stack*           ls;         // one lane per path (YH_LANE, dev_lane.h): this lane's stack, else unused
  const yhd_scene*      sc_dev;     // YH_LANE: a copy of *sc in device memory, for out-of-line callees (the kernel
                                    // argument itself must not have its address escape: it would be copied to scratch)
};
// Stages the tables every kernel keeps in LDS — the scene level (objects, scene BVH nodes and primitives; when it
// fits), the camera, the small area lights and the environment cdf index — at `at` (YHD_LDS_TABLES_F4 float4) and
// points `tc` at them. All threads of the block call it; a __syncthreads() must follow.
YH_DEV void stage_tables(const yhd_scene& sc, YH_LDS v4f* at, int tid, int nthreads, trace_ctx& tc, YH_LDS float*& lds_cam) {
  tc.lds_scene = nullptr, tc.lds_lights = nullptr, tc.lds_envtab = nullptr, tc.lds_mats = nullptr;
  if (sc.lds_scene_f4 > 0) {
    const int  nobj = YH_OBJECT_F4 * sc.num_objects, nnod = 2 * sc.num_scene_nodes, npri = (sc.num_objects + 3) / 4;
    const v4f* gobj = (const v4f*)sc.objects;
    const v4f* gpri = (const v4f*)sc.scene_prims;  // padded to a multiple of 4 ints by the host
    for (int i = tid; i < nobj; i += nthreads) at[i] = gobj[i];
    for (int i = tid; i < nnod; i += nthreads) at[nobj + i] = *(const v4f*)(sc.scene_nodes + i);
    for (int i = tid; i < npri; i += nthreads) at[nobj + nnod + i] = gpri[i];
    tc.lds_scene = at;
  }
  // the camera too: as kernel arguments its 17 floats end up in spilled SGPRs reloaded one v_readlane at a time
  lds_cam = (YH_LDS float*)(at + sc.lds_scene_f4);
  if (tid < 17) lds_cam[tid] = ((const float*)&sc.camera)[tid];
  YH_LDS v4f* lights = at + sc.lds_scene_f4 + 5;
  if (sc.light_table_f4 > 0) {
    for (int i = tid; i < sc.light_table_f4; i += nthreads) lights[i] = *(const v4f*)(sc.light_table + i);
    tc.lds_lights = lights;
  }
  YH_LDS float* envtab = (YH_LDS float*)(lights + sc.light_table_f4);
  if (sc.env_tab_k > 0) {
    for (int i = tid; i < sc.env_tab_k; i += nthreads) envtab[i] = sc.env_tab[i];
    tc.lds_envtab = envtab;
  }
  // the material table: per-material hair constants and surface parameters, read by every shaded hit
  YH_LDS v4f* mats = (YH_LDS v4f*)(envtab + (sc.env_tab_k + 3) / 4 * 4);
  if (sc.lds_materials > 0) {
    const v4f* gm = (const v4f*)sc.materials;
    for (int i = tid; i < YH_MATERIAL_F4 * sc.lds_materials; i += nthreads) mats[i] = gm[i];
    tc.lds_mats = mats;
  }
}

// Per-lane counters of the instrumented build (COUNT = true): kept in registers
// for a whole work item and flushed once, so that the instrumented kernel runs
// at nearly the speed of the production one and its cycle stamps mean something.
struct stats_t {
  unsigned int samples, rays, nodes, seg, tri, hair, surf, envl, envs;
  unsigned long long c_geom, c_sample, c_eval, c_rest;  // shader-clock stamps inside path_step
  // divergence profile of the traversal loop: for each kind of step, how many wave trips
  // executed its code (t_*) and how many lanes were active in it (l_*)
  unsigned int t_node, l_node, t_line, l_line, t_tri, l_tri, t_enter, l_enter, t_scene, l_scene;
};
// Inside a divergent branch: one count per wave trip + the active lanes of that trip.
template <bool COUNT>
YH_DEV void count_branch(unsigned int& trips, unsigned int& lanes) {
  if (COUNT) {
    unsigned long long m = __ballot(1);
    if ((int)__lane_id() == __ffsll((long long)m) - 1) trips++, lanes += (unsigned int)__popcll(m);
  }
}
// one 16-byte load (never split into dwordx3 + dword)
YH_DEV v4f ldg4(const yhd_float4* p) { return *(const v4f*)p; }
YH_DEV f3  xyz(v4f a) { return f3{a.x, a.y, a.z}; }

// one count per quad (the four lanes of a quad run the same path)
template <bool COUNT>
YH_DEV void count_quad(unsigned int& slot) {
  if (COUNT && (__lane_id() & 3u) == 0) slot++;
}

// Closest hit against the whole scene (first_object < 0) or against a single
// instance (intersect_instance_bvh, pt.cpp:1031-1037). Called by all four
// lanes of a quad with identical arguments; all four return the same hit.
//
// One node step = one 128-byte fetch (32 bytes per lane) + one of the
// reference's slab tests per lane; the hit children are
// visited in exactly the order the reference's binary traversal visits them
// (near side first by the sign of the ray direction on each split axis,
// pt.cpp:887-893), so `tmax` shrinks identically and exact-t ties resolve
// identically. One leaf step = one primitive test per lane, then the
// reference's sequential accept rule (math.h:3450: reject only t > tmax, so
// among equal t the LATER primitive wins) applied as a quad reduction.
// EXACT = false: box tests use intersect_bbox_nonan; a lane whose ray could put a NaN into a slab
// (non-finite origin or 1 / d, in world space or in the space of an object it enters) stops and
// reports `redo`, and trace_ray repeats that ray with EXACT = true, the reference's compare +
// select form throughout. Results are identical either way; only axis-parallel rays take the
// second pass.
// Nodes and leaf test records are read from yhd_scene::lane_blob (yh_device.h: ONE array in 32-byte units with absolute references —
// the 4-wide nodes and the 8- / 16-wide ones behind them, all made at yh_upload_scene): one base and one address form for
// whatever a quad, an octet or a group of sixteen holds. (Until round 4 the kernels read `nodes` / `nodes8` / `nodes16` + `prims`;
// the A/B halves, the resumable PHASE form, the cache-line touches of pushed children and the quad form over 8-wide nodes are
// closed experiments: profiles/r05/pruned_experiments.patch holds their code, profiles/r02-r04 their numbers.)
// LDS_SCENE = true: the caller knows the scene-level table is in LDS (the plain kernel variants: the host selects the
// GENERAL ones when it does not fit), so the loop carries no second code path for reading it from memory.
// MODE: how many lanes own the ray and how many binary levels a node step covers.
//   YH_MODE_QUAD  four lanes, 4-wide nodes (two levels): lane q tests slot q.
//   YH_MODE_OCT   EIGHT lanes, 8-wide nodes: lane o tests slot o; the two quads of the octet hold the same path and run
//                 everything else (scene level, ENTER, leaves, shading) twice over. Eight paths share a wave instead of
//                 sixteen: for launches bound by the chain of steps of ONE path (few expensive pixels per GPU).
// The children of a wide node are visited in the reference's order in every mode (ranks below), so closest hits,
// `tmax` and exact-t ties are the same.
#define YH_MODE_QUAD 0
#define YH_MODE_OCT 2
#define YH_MODE_HEX 3 /* SIXTEEN lanes, 16-wide nodes (four levels, host/bvh_build.h: WideNode16): lane o tests slot o; the four quads hold the same path */
#define YH_ROW_HALF_MIRROR 0x141 /* DPP: lane i of every eight reads lane 7 - i */
#define YH_ROW_MIRROR 0x140      /* DPP: lane i of every sixteen reads lane 15 - i */
#define YH_MODE_OCTP 4 /* YH_MODE_OCT with LEAF PAIRS: the octet's two quads test two leaves in one step (leaf groups, below). Bit-identical; 6 % fewer trips and
                          2-6 % slower on C1 (the look at the stack's top and the exchange between the quads cost every leaf step), 17 % FASTER on hair-curls
                          (many leaf steps per ray): its own launch shape, the trials decide (profiles/r03/oct_leaf_pairs_ab.txt, hex_ab.txt) */
#define YH_IS_OCT(MODE) ((MODE) == YH_MODE_OCT || (MODE) == YH_MODE_OCTP)
#define YH_MODE_HEXP 5 /* YH_MODE_HEX with LEAF GROUPS: the four quads of the sixteen test up to four leaves in one step (the entry and the leaves on top of it
                          on the stack); the merge keeps the reference's order. C1 at 180^2 +5-8 %, hair-curls at 320^2 1.47 x over YH_MODE_HEX, `textured` equal
                          (profiles/r03/hex_leaf_groups_ab.txt). Its own launch shape (8), the trials decide */
#define YH_IS_HEX(MODE) ((MODE) == YH_MODE_HEX || (MODE) == YH_MODE_HEXP)
template <bool COUNT, int STRIDE, bool EXACT, bool LDS_SCENE = false, int MODE = YH_MODE_QUAD>
YH_DEV hit_t trace_ray_loop(const trace_ctx& tc, const ray_t& ray, int first_object, unsigned int* steps_out, bool& redo) {
  const yhd_scene&     sc   = *tc.sc;
  const unsigned int   q    = __lane_id() & 3u;
  const unsigned int   q_   = q;
  int                  sp   = 0;
  YH_LDS unsigned int* hrec = tc.lds_stack;  // rows 0-2 of the column: the surviving hit's u, v (raw), r; the stack proper starts at row YH_HITROWS
#define YH_STK(i) hrec[((i) + YH_HITROWS) * STRIDE] /* (one base pointer for both: the row offset folds into the LDS instruction) */
  auto push = [&](unsigned int v) { YH_STK(sp) = v, sp++; };
  auto pop  = [&]() -> unsigned int { sp--; return YH_STK(sp); };
  hit_t hit;
  hit.object = -1, hit.slot = -1, hit.u = 0, hit.v = 0, hit.distance = 0;
  float tmax = ray.tmax;
  // world-space ray data for the scene level
  f3  wdinv = quad_rcp(ray.d);  // one division per lane of the quad (dev_math.h)
  const int wsign = (wdinv.x < 0 ? 1 : 0) | (wdinv.y < 0 ? 2 : 0) | (wdinv.z < 0 ? 4 : 0);
  const bool wnonan = finite3(wdinv) && finite3(ray.o);
  auto box_test = [](f3 o, f3 dinv, float t0, float t1, f3 bmin, f3 bmax) {
    return EXACT ? intersect_bbox(o, dinv, t0, t1, bmin, bmax) : intersect_bbox_nonan(o, dinv, t0, t1, bmin, bmax);
  };
  if (!EXACT && !wnonan) {
    redo = true;
    return hit;
  }
  // instance-space ray data
  f3    lo = ray.o, ld = ray.d, ldinv = wdinv;
  // dot(ld, ld): the `a` of every line test inside the entered object (math.h:3437), set at ENTER — except in the dense shape (96
  // registers: one more live value puts a spill reload into the loop, tools/check_codegen.py), which computes it per test
  constexpr bool HOIST_A = STRIDE != 64;
  float ld2 = 0.0f;
  // sign bits of 1 / d (x | y << 1 | z << 2) of the ray in the entered object's space in bits 0-2 and of the world-space ray in bits
  // 4-6 of ONE register (a split axis is 0, 1 or 2, so `(lsign >> axis) & 1` reads the right bit either way)
  int   lsign = wsign | (wsign << 4), cur_obj = -1, kind = 0, node_base = 0;
  unsigned long long n_nodes = 0, n_seg = 0, n_tri = 0;
  unsigned int       n_steps = 0;

  const bool in_lds = LDS_SCENE || tc.lds_scene != nullptr;
  const YH_LDS v4f* lds_snodes = in_lds ? tc.lds_scene + YH_OBJECT_F4 * sc.num_objects : nullptr;
  auto scene_prim = [&](int i) -> int {
    if (in_lds) return ((const YH_LDS int*)(lds_snodes + 2 * sc.num_scene_nodes))[i];
    return sc.scene_prims[i];
  };
  unsigned int cur;
  if (first_object >= 0) {
    cur = YH_TAG_ENTER | (unsigned)first_object;
  } else {
    if (sc.num_scene_nodes == 0) return hit;
    cur = YH_TAG_SCENE | 0u;
  }
  // One loop, one dependent fetch per iteration: whatever a quad holds — a wide
  // node or a leaf — every lane fetches "its" 32 bytes (slot q of the node, or
  // the test half of primitive q of the leaf) with the same two load
  // instructions, and only the arithmetic diverges. The wave therefore pays one
  // memory round trip per step of its slowest quad, not one per kind of step.
  while (true) {
    if (COUNT) n_steps++;
    if (cur == YH_NONE) {
      if (sp == 0) break;
      cur = pop();
    }
    unsigned int tag = cur & YH_TAG_MASK;
    // The scene level costs no trip of its own: a scene node (read from the LDS copy
    // of the tiny scene BVH), the ENTER it leads to and the root fetch of the entered
    // shape chain inside ONE iteration, so a ray pays one memory round trip per
    // object it enters instead of three.
    if (tag == YH_TAG_SCENE) {
      // scene-level node (binary, reference layout; every lane of the quad does it)
      if (COUNT) count_branch<COUNT>(tc.stats->t_scene, tc.stats->l_scene);
      int idx = (int)(cur & ~YH_TAG_MASK);
      v4f n0, n1;
      if (in_lds) n0 = lds_snodes[2 * idx], n1 = lds_snodes[2 * idx + 1];
      else n0 = ldg4(sc.scene_nodes + 2 * idx), n1 = ldg4(sc.scene_nodes + 2 * idx + 1);
      if (q == 0) n_nodes++;
      cur = YH_NONE;
      if (box_test(ray.o, wdinv, ray.tmin, tmax, xyz(n0), xyz(n1))) {
        int start = __float_as_int(n0.w), meta = __float_as_int(n1.w);
        if (meta & 0x10000) {  // internal
          int axis = (meta >> 24) & 3;
          int near = (lsign >> (4 + axis)) & 1;  // dsign set: visit start+1 first
          push(YH_TAG_SCENE | (unsigned)(start + 1 - near));
          cur = YH_TAG_SCENE | (unsigned)(start + near);
        } else {
          // a leaf of the scene BVH: its objects are entered in order, i.e. pushed in reverse behind the first (pt.cpp:1005-1023). A
          // leaf of the reference's tree holds at most four (pt.cpp:598), so lane q of the quad pushes object q — one predicated
          // LDS write, no loop
          int num = meta & 0xffff;
          if ((int)q >= 1 && (int)q < num) YH_STK(sp + (num - 1 - (int)q)) = YH_TAG_ENTER | (unsigned)scene_prim(start + (int)q);
          sp += num > 1 ? num - 1 : 0;
          if (num > 0) cur = YH_TAG_ENTER | (unsigned)scene_prim(start);
        }
      }
      tag = cur & YH_TAG_MASK;
      if (cur == YH_NONE || tag == YH_TAG_SCENE) continue;
    }
    if (tag == YH_TAG_ENTER) {
      // transform_ray(inverse(object.frame, true), ray) (pt.cpp:1012-1013)
      if (COUNT) count_branch<COUNT>(tc.stats->t_enter, tc.stats->l_enter);
      cur_obj             = (int)(cur & ~YH_TAG_MASK);
      // A ray that misses the object's padded world box cannot hit anything in it: the reference
      // would enter, test the shape's root box in object space and leave (pt.cpp:1012-1016); the
      // result is the same without the inverse transform, the three divisions and the root fetch.
      // Skipped for axis-parallel rays (a 0 * inf slab would make the test inconclusive).
      if (wnonan) {
        v4f bmin, bmax;
        if (in_lds) {
          const YH_LDS v4f* ob = tc.lds_scene + YH_OBJECT_F4 * cur_obj;
          bmin = ob[8], bmax = ob[9];
        } else {
          const yhd_object& o = sc.objects[cur_obj];
          bmin = v4f{o.wbox_min[0], o.wbox_min[1], o.wbox_min[2], 0}, bmax = v4f{o.wbox_max[0], o.wbox_max[1], o.wbox_max[2], 0};
        }
        if (!box_test(ray.o, wdinv, ray.tmin, tmax, xyz(bmin), xyz(bmax))) {
          cur = YH_NONE;
          continue;
        }
      }
      frame inv;
      if (in_lds) {  // yhd_object: frame[12] inv_frame[12] kind ... lane_root lane_test lane_root8 lane_root16 (11 float4)
        const YH_LDS v4f* ob = tc.lds_scene + YH_OBJECT_F4 * cur_obj;
        v4f a = ob[3], b = ob[4], c = ob[5], d = ob[6];
        inv.x = {a.x, a.y, a.z}, inv.y = {a.w, b.x, b.y}, inv.z = {b.z, b.w, c.x}, inv.o = {c.y, c.z, c.w};
        kind = __float_as_int(d.x);
        // the shape's root in the lane blob, 32-byte units: yhd_object::lane_root / lane_root8 / lane_root16
        node_base = __float_as_int(YH_IS_HEX(MODE) ? ob[10].w : YH_IS_OCT(MODE) ? ob[10].z : ob[10].x);
      } else {
        const yhd_object& o = sc.objects[cur_obj];
        inv       = ldframe(o.inv_frame);
        kind      = o.kind;
        node_base = YH_IS_HEX(MODE) ? o.lane_root16 : YH_IS_OCT(MODE) ? o.lane_root8 : o.lane_root;
      }
      lo    = transform_point(inv, ray.o);
      ld    = transform_vector(inv, ray.d);
      ldinv = quad_rcp(ld);
      lsign = (lsign & 0x70) | (ldinv.x < 0 ? 1 : 0) | (ldinv.y < 0 ? 2 : 0) | (ldinv.z < 0 ? 4 : 0);
      if (!EXACT && !(finite3(ldinv) && finite3(lo))) {  // a slab of this object could hold a NaN: second pass
        redo = true;
        break;
      }
      if (HOIST_A) ld2 = dot(ld, ld);
      cur = YH_TAG_SHAPE | (unsigned)node_base;  // shape root: fetched in this same iteration
      tag = YH_TAG_SHAPE;
    }
    {
      bool is_leaf    = tag == YH_TAG_LEAF;
      // LEAF GROUPS (YH_MODE_OCTP: pairs, YH_MODE_HEXP: up to four): when the entry is a leaf and the entries on top of the
      // stack are leaves too (siblings that were both hit: the common case in hair), the other quads of the path's lanes
      // test those in the same step: quad j (j = 1..) looks at the j-th entry below the stack's top; the first k of them
      // that are leaves (consecutive from the top) are tested next to the entry, quad j testing the one that would be
      // popped j-th. The primitive test depends on `tmax` only through its final `t > tmax` reject, so testing against
      // the older, longer `tmax` accepts a superset, and the merge below picks what the sequential order keeps.
      constexpr bool     GROUPS = MODE == YH_MODE_OCTP || MODE == YH_MODE_HEXP;
      const unsigned int qj     = MODE == YH_MODE_HEXP ? (__lane_id() >> 2) & 3u : MODE == YH_MODE_OCTP ? (__lane_id() >> 2) & 1u : 0u;
      int                grp    = 0;      // k
      bool               idle   = false;  // a quad beyond the group: tests the entry again, its result is dropped
      unsigned int       mycur  = cur;    // the leaf this lane's quad tests
      if (GROUPS) {
        const bool   look = is_leaf && qj > 0 && sp >= (int)qj;  // (YH_NONE itself carries the leaf tag: an empty stack must not read as a leaf)
        unsigned int pk   = look ? YH_STK(sp - (int)qj) : 0u;
        unsigned int lb   = (look && (pk & YH_TAG_MASK) == YH_TAG_LEAF) ? (1u << qj) : 0u;
        lb |= (unsigned int)dpp_i<YH_ROW_HALF_MIRROR>((int)lb);
        if (MODE == YH_MODE_HEXP) lb |= (unsigned int)dpp_i<YH_ROW_MIRROR>((int)lb);
        grp   = (lb & 2u) ? ((lb & 4u) ? ((lb & 8u) ? 3 : 2) : 1) : 0;
        idle  = is_leaf && (int)qj > grp;
        mycur = (qj > 0 && !idle) ? pk : cur;
      }
      int  leaf_start = (int)(mycur & 0x07FFFFFFu), leaf_num = (int)((mycur >> 27) & 7u);
      bool mine       = !is_leaf || (int)q < leaf_num;  // lanes beyond the leaf's count re-read its last record
      int  pq         = mine ? (int)q : leaf_num - 1;
      // the lane's 32 bytes in the blob: slot (lane of the group) of the node, or the test record of primitive pq of the leaf (leaf_start: its first test record)
      const unsigned int nslot = MODE == YH_MODE_QUAD ? q : (__lane_id() & (YH_IS_HEX(MODE) ? 15u : 7u));
      const yhd_float4*  addr  = sc.lane_blob + 2 * (size_t)(is_leaf ? (unsigned)leaf_start + (unsigned)pq * (kind == YH_KIND_LINES ? 1u : 2u) : cur + nslot);
      if (GROUPS) mine = mine && !idle;
      const v4f s0 = ldg4(addr), s1 = ldg4(addr + 1);
      if (YH_IS_HEX(MODE) && !is_leaf) {
        // ---- 16-wide node: slot o = s1 << 3 | s2 << 2 | s3 << 1 | s4, one per lane of the sixteen; the rank of a slot in the
        // reference's visiting order is the four near / far decisions of pt.cpp:887-893 at the four collapsed levels.
        if (q == 0) n_nodes++;
        if (COUNT) count_branch<COUNT>(tc.stats->t_node, tc.stats->l_node);
        const unsigned int o    = __lane_id() & 15u;
        const unsigned int axes = __float_as_uint(s1.w);
        const unsigned int n0   = (lsign >> (axes & 3)) & 1;
        const unsigned int n1   = (lsign >> ((axes >> (2 + 2 * (o >> 3))) & 3)) & 1;
        const unsigned int n2   = (lsign >> ((axes >> (6 + 2 * (o >> 2))) & 3)) & 1;
        const unsigned int n3   = (lsign >> ((axes >> (14 + 2 * (o >> 1))) & 3)) & 1;
        const unsigned int rank = (((o >> 3) ^ n0) << 3) | ((((o >> 2) & 1) ^ n1) << 2) | ((((o >> 1) & 1) ^ n2) << 1) | ((o & 1) ^ n3);
        unsigned int ref = __float_as_uint(s1.z);
        const bool   h   = box_test(lo, ldinv, ray.tmin, tmax, f3{s0.x, s0.y, s0.z}, f3{s0.w, s1.x, s1.y}) && ref != YH_NONE;
        unsigned int bits = h ? (1u << rank) : 0u;
        unsigned int M    = bits | (unsigned int)dpp_i<YH_QUAD_XOR1>((int)bits);
        M |= (unsigned int)dpp_i<YH_QUAD_XOR2>((int)M);
        M |= (unsigned int)dpp_i<YH_ROW_HALF_MIRROR>((int)M);
        M |= (unsigned int)dpp_i<YH_ROW_MIRROR>((int)M);  // hits in visiting order, bit k = k-th visited
        const bool first = h && (1u << rank) == (M & (0u - M));
        if (h && !first) YH_STK(sp + (int)__popc(M >> (rank + 1))) = ref;
        unsigned int mine = first ? ref : 0u;
        mine |= (unsigned int)dpp_i<YH_QUAD_XOR1>((int)mine);
        mine |= (unsigned int)dpp_i<YH_QUAD_XOR2>((int)mine);
        mine |= (unsigned int)dpp_i<YH_ROW_HALF_MIRROR>((int)mine);
        mine |= (unsigned int)dpp_i<YH_ROW_MIRROR>((int)mine);
        int nh = __popc(M);
        sp += nh > 0 ? nh - 1 : 0;
        cur = nh > 0 ? mine : YH_NONE;
      } else if (MODE != YH_MODE_QUAD && !is_leaf) {
        // ---- 8-wide node: slot o = s1 << 2 | s2 << 1 | s3 (host/bvh_build.h), one per lane of the octet. The visiting order applies
        // pt.cpp:887-893 at the three collapsed levels: rank bit 2 = side of the node's own axis, bit 1 = side of the child's, bit 0 =
        // side of the grandchild's, each flipped when the ray runs against that axis. Every hit slot pushes itself so
        // that the stack pops in visiting order; the first visited becomes `cur`.
        if (q == 0) n_nodes++;
        if (COUNT) count_branch<COUNT>(tc.stats->t_node, tc.stats->l_node);
        const unsigned int axes = __float_as_uint(s1.w);
        const unsigned int g    = (__lane_id() & 7u) >> 1;  // this lane's grandchild: 2 * s1 + s2
        const unsigned int n0   = (lsign >> (axes & 3)) & 1;
        const unsigned int n1   = (lsign >> ((axes >> (2 + 2 * (g >> 1))) & 3)) & 1;
        const unsigned int n2   = (lsign >> ((axes >> (6 + 2 * g)) & 3)) & 1;
        const unsigned int rank = ((((g >> 1) ^ n0) << 2) | (((g & 1) ^ n1) << 1)) | ((__lane_id() & 1u) ^ n2);
        const unsigned int ref  = __float_as_uint(s1.z);
        const bool         h    = box_test(lo, ldinv, ray.tmin, tmax, f3{s0.x, s0.y, s0.z}, f3{s0.w, s1.x, s1.y}) && ref != YH_NONE;
        unsigned int bits = h ? (1u << rank) : 0u;
        unsigned int M    = bits | (unsigned int)dpp_i<YH_QUAD_XOR1>((int)bits);
        M |= (unsigned int)dpp_i<YH_QUAD_XOR2>((int)M);
        M |= (unsigned int)dpp_i<YH_ROW_HALF_MIRROR>((int)M);  // hits in visiting order, bit k = k-th visited
        const bool first = h && (1u << rank) == (M & (0u - M));  // the first visited hit slot
        if (h && !first) YH_STK(sp + (int)__popc(M >> (rank + 1))) = ref;
        unsigned int mine = first ? ref : 0u;  // child refs are never 0 (a node's offset lies behind the test records)
        mine |= (unsigned int)dpp_i<YH_QUAD_XOR1>((int)mine);
        mine |= (unsigned int)dpp_i<YH_QUAD_XOR2>((int)mine);
        mine |= (unsigned int)dpp_i<YH_ROW_HALF_MIRROR>((int)mine);
        int nh = __popc(M);
        sp += nh > 0 ? nh - 1 : 0;
        cur = nh > 0 ? mine : YH_NONE;
      } else if (!is_leaf) {
        // ---- wide node: lane q tests slot q {min.xyz, max.x} {max.yz, ref, axes} ----
        if (q == 0) n_nodes++;
        if (COUNT) count_branch<COUNT>(tc.stats->t_node, tc.stats->l_node);
        bool h = box_test(lo, ldinv, ray.tmin, tmax, f3{s0.x, s0.y, s0.z}, f3{s0.w, s1.x, s1.y});
        unsigned int ref  = __float_as_uint(s1.z);
        unsigned int axes = __float_as_uint(s1.w);
        h = h && ref != YH_NONE;  // an empty slot's inverted box still passes the min/max slab test
        // Visiting order of the four slots (pt.cpp:887-893 applied at both collapsed
        // levels): the pair on the near side of the node's own axis first, and
        // inside each pair the slot on the near side of that child's axis first.
        // Each lane computes the RANK of its own slot in that order; the hit lanes
        // then push themselves in one parallel step: the first hit in visiting order
        // becomes `cur`, the others go on the stack so that they pop in visiting order.
        unsigned int q = q_;
        if (STRIDE == 64) {  // dense shape, 96 registers: lane & 3 recomputed here (two instructions) instead of a lane constant that the allocator reloads from scratch
          unsigned int x;
          asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
          q = x & 3u;
        }
        unsigned int pair = q >> 1;
        unsigned int sgn  = (lsign >> ((axes >> (2 + 2 * pair)) & 3)) & 1;   // near side of this pair's own axis
        unsigned int s0_  = (lsign >> (axes & 3)) & 1;                      // near side of the node's axis
        unsigned int rank = ((pair ^ s0_) << 1) | ((q & 1) ^ sgn);
        unsigned int bit  = h ? (1u << rank) : 0u;
        unsigned int M    = bit | (unsigned int)dpp_i<YH_QUAD_XOR1>((int)bit);
        M |= (unsigned int)dpp_i<YH_QUAD_XOR2>((int)M);                     // hits in visiting order, bit k = k-th visited
        bool         first = h && (M & (bit - 1)) == 0;
        unsigned int after = (unsigned int)__popc(M >> (rank + 1));         // hit slots visited after this one
        if (h && !first) YH_STK(sp + (int)after) = ref;
        unsigned int mine = first ? ref : 0u;                                // child refs are never 0 (node 0 is a root)
        mine |= (unsigned int)dpp_i<YH_QUAD_XOR1>((int)mine);
        mine |= (unsigned int)dpp_i<YH_QUAD_XOR2>((int)mine);
        int nh = __popc(M);
        sp += nh > 0 ? nh - 1 : 0;
        cur = nh > 0 ? mine : YH_NONE;
      } else {
        // ---- leaf: lane q tests primitive q, in leaf order (pt.cpp:905-923) ---------
        cur = YH_NONE;
        bool  ok = false;
        float uu = 0, vv = 0, dist = 0, rr = 1.0f;
        if (kind == YH_KIND_LINES) {
          if (COUNT) count_branch<COUNT>(tc.stats->t_line, tc.stats->l_line);
          if (mine) {
            n_seg++;
            ok = intersect_line<(STRIDE == 64), true>(lo, ld, ray.tmin, tmax, xyz(s0), xyz(s1), s0.w, s1.w, uu, vv, dist, HOIST_A ? ld2 : dot(ld, ld), &rr);  // dense shape: 256 threads = 64 quads
          }
        } else {
          if (COUNT) count_branch<COUNT>(tc.stats->t_tri, tc.stats->l_tri);
          v4f s2 = ldg4(addr + 2);
          if (mine) {
            n_tri++;
            ok = intersect_triangle(lo, ld, ray.tmin, tmax, xyz(s0), xyz(s1), xyz(s2), uu, vv, dist);
          }
        }
        // The reference tests the leaf's primitives in order, shrinking tmax after
        // each accepted hit: the survivor is the accepted primitive of minimum t,
        // the LATER one among equal t. Same result as a quad min-reduction.
        const int my_i  = (int)(4 * qj + q);  // (qj: the quad's place in a leaf group, 0 in the modes without)
        int       key_i = ok ? my_i : -1;
        float     key_t = dist;
        int       slot  = leaf_start + (int)q * (kind == YH_KIND_LINES ? 1 : 2);  // this lane's primitive: its test record in the blob
#define YH_KEY_MERGE(CTRL, WITH_SLOT)                                                      \
  {                                                                                        \
    int   oi = dpp_i<CTRL>(key_i);                                                         \
    float ot = dpp_f<CTRL>(key_t);                                                         \
    bool  take = (oi >= 0) & ((key_i < 0) | (ot < key_t) | ((ot == key_t) & (oi > key_i))); /* no short circuits: selects, not branches */ \
    if (WITH_SLOT) {                                                                       \
      int os = dpp_i<CTRL>(slot);                                                          \
      slot   = take ? os : slot;                                                           \
    }                                                                                      \
    key_i = take ? oi : key_i, key_t = take ? ot : key_t;                                  \
  }
        YH_KEY_MERGE(YH_QUAD_XOR1, false)
        YH_KEY_MERGE(YH_QUAD_XOR2, false)
        if (GROUPS) {
          // The reference meets the group's leaves one after the other, each against the ray shortened by the hits before
          // it: the survivor is the accepted primitive of minimum t, the LATER one among equal t (math.h:3450) — the same
          // rule as inside a leaf, so the merge simply goes on across the quads with the leaf's place in the key.
          slot = leaf_start + (key_i & 3) * (kind == YH_KIND_LINES ? 1 : 2);
          YH_KEY_MERGE(YH_ROW_HALF_MIRROR, true)
          if (MODE == YH_MODE_HEXP) YH_KEY_MERGE(YH_ROW_MIRROR, true)
          sp -= grp;  // the group's leaves came off the stack
        } else {
          slot = leaf_start + (key_i & 3) * (kind == YH_KIND_LINES ? 1 : 2);
        }
#undef YH_KEY_MERGE
        if (key_i >= 0) {
          hit.object = cur_obj, hit.slot = slot;  // (the primitive's test record; made the leaf-order index below)
          hit.distance = key_t;
          tmax = key_t;
          if (key_i == my_i) hrec[0] = __float_as_uint(uu), hrec[STRIDE] = __float_as_uint(vv), hrec[2 * STRIDE] = __float_as_uint(rr);  // the survivor's lane (the octet / sixteen forms run the same path in every quad: any one of the identical writers)
        }
      }
    }
  }
  if (hit.object >= 0) {  // test record in the blob -> leaf-order index of the primitive in its shape (hit_t)
    int hk, lt;
    if (in_lds) {
      const YH_LDS v4f* ob = tc.lds_scene + YH_OBJECT_F4 * hit.object;
      hk = __float_as_int(ob[6].x), lt = __float_as_int(ob[10].y);
    } else {
      hk = sc.objects[hit.object].kind, lt = sc.objects[hit.object].lane_test;
    }
    hit.slot = hk == YH_KIND_LINES ? hit.slot - lt : (hit.slot - lt) >> 1;
    {  // the survivor's u, v from the hit record; uv.y = sqrt(d2) / r of a line (math.h:3465) evaluated here, once
      const float hu = __uint_as_float(hrec[0]), hv = __uint_as_float(hrec[STRIDE]), hr = __uint_as_float(hrec[2 * STRIDE]);
      hit.u = hu, hit.v = hk == YH_KIND_LINES ? sqrtf(hv) / hr : hv;
    }
  }
  if (COUNT) {
    if (steps_out) *steps_out = n_steps;
    tc.stats->nodes += (unsigned int)n_nodes, tc.stats->seg += (unsigned int)n_seg, tc.stats->tri += (unsigned int)n_tri;
  }
  return hit;
}
#undef YH_STK

template <bool COUNT, int STRIDE, bool LDS_SCENE = false, int MODE = YH_MODE_QUAD>
YH_DEV hit_t trace_ray(const trace_ctx& tc, const ray_t& ray, int first_object, unsigned int* steps_out = nullptr) {
  bool  redo = false;
  hit_t hit  = trace_ray_loop<COUNT, STRIDE, false, LDS_SCENE, MODE>(tc, ray, first_object, steps_out, redo);
  if (__any(redo)) {
    if (redo) hit = trace_ray_loop<COUNT, STRIDE, true, LDS_SCENE, MODE>(tc, ray, first_object, steps_out, redo);
  }
  return hit;
}

// Element id (the reference's `element`) of a hit, read from its leaf record.
YH_DEV int hit_element(const yhd_scene& sc, const hit_t& hit) {
  if (hit.object < 0) return -1;
  const yhd_object& o = sc.objects[hit.object];
  if (o.kind == YH_KIND_LINES) return __float_as_int(sc.prims[(size_t)o.prim_base + (size_t)hit.slot * 4 + 2].w);
  return __float_as_int(sc.prims[(size_t)o.prim_base + (size_t)hit.slot * 6].w);
}

}  // namespace yhd
#endif
