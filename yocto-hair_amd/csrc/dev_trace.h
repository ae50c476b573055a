// dev_trace.h — PCG32, ray / primitive tests and the two-level BVH traversal.
//
// Restates math.h:1396-1442 (PCG32), math.h:3426-3505,3544-3554
// (intersect_line / _triangle / _bbox) and pt.cpp:821-1053
// (intersect_shape_bvh / intersect_scene_bvh / intersect_instance_bvh).
//
// MI355X-first differences from the reference's nested, pointer-chasing
// loops (results are unchanged):
//   * ONE flattened traversal loop with one stack for both levels: stack
//     entries are tagged {scene node, enter-instance, shape node}; the
//     instance's leaf loop `for idx in start..start+num` of the scene level
//     becomes enter-instance entries pushed in reverse order, so instances
//     are visited in the reference's order and `tmax` shrinks identically.
//   * inverse object frames are precomputed at upload instead of per ray per
//     object (pt.cpp:1012-1013 recomputes a 3x3 adjugate inverse each time);
//   * leaf primitives are 16-byte-aligned records in leaf order (yh_device.h)
//     so a segment test costs two dwordx4 loads and no index indirection;
//   * BVH nodes whose index falls in the LDS-resident window (the top of the
//     hair tree) are read from LDS.
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
YH_DEV bool intersect_line(f3 ro, f3 rd, float tmin, float tmax, f3 p0, f3 p1, float r0,
    float r1, float& uu, float& vv, float& dist) {
  f3    u = rd, v = p1 - p0, w = ro - p0;
  float a = dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w);
  float det = a * c - b * b;
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
  uu = s, vv = sqrtf(d2) / r;
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

// ---------------------------------------------------------------------------
// Traversal
// ---------------------------------------------------------------------------
#define YH_TAG_SHAPE 0u          /* entry = global wide-node index            */
#define YH_TAG_SCENE 0x40000000u /* entry = scene-node index                 */
#define YH_TAG_ENTER 0x80000000u /* entry = object id to enter               */
#define YH_TAG_LEAF 0xC0000000u  /* entry = count << 27 | first leaf slot    */
#define YH_TAG_MASK 0xC0000000u
#define YH_NONE 0xFFFFFFFFu

// Traversal stack. In k_trace the first YH_LDS_STACK entries of every lane
// live in LDS (column `tid` of a [depth][block] array: conflict-free, one
// ds_write_b32 / ds_read_b32 per push / pop) and only deeper entries overflow
// to scratch; kernels without an LDS carve-out (unit-level batches) use
// scratch only. Pointers into LDS carry the LDS address space so that the
// compiler emits ds_* instructions instead of flat ones.
#ifndef YH_LDS_STACK
#define YH_LDS_STACK 24
#endif
#define YH_LDS __attribute__((address_space(3)))
typedef float v4f __attribute__((ext_vector_type(4)));

struct trace_ctx {
  const yhd_scene*      sc;
  const YH_LDS v4f*     lds_nodes;  // LDS copy of nodes[lds_node_base ..+count)
  YH_LDS unsigned int*  lds_stack;  // this lane's LDS stack column
  yhd_counters*         counters;   // NULL in the production kernel
};
// one 16-byte load (never split into dwordx3 + dword)
YH_DEV v4f ldg4(const yhd_float4* p) { return *(const v4f*)p; }
YH_DEV f3  xyz(v4f a) { return f3{a.x, a.y, a.z}; }

template <bool COUNT>
YH_DEV void count_add(unsigned long long* slot, unsigned long long n) {
  if (COUNT) atomicAdd(slot, n);
}

// Closest hit against the whole scene (first_object < 0) or against a single
// instance (intersect_instance_bvh, pt.cpp:1031-1037).
//
// "while-while" traversal over the 4-wide tree: every lane first walks nodes
// until it owns a leaf (or has nothing left), then the lanes of the wave test
// their leaf's primitives together. One node step = one 128-byte fetch + four
// of the reference's slab tests; the hit children are visited in exactly the
// order the reference's binary traversal visits them (near side first by the
// sign of the ray direction on each split axis, pt.cpp:887-893), so `tmax`
// shrinks identically and exact-t ties resolve identically.
template <bool COUNT, bool LDS, int STRIDE>
YH_DEV hit_t trace_ray(const trace_ctx& tc, const ray_t& ray, int first_object, unsigned int* steps_out = nullptr) {
  const yhd_scene& sc = *tc.sc;
  // stack: `sp` and the LDS column pointer stay in registers; only the
  // overflow array is addressable memory
  constexpr int         kLds = LDS ? YH_LDS_STACK : 0;
  unsigned int          ovf[YH_STACK_MAX - kLds];
  int                   sp   = 0;
  YH_LDS unsigned int*  lstk = tc.lds_stack;
  auto push = [&](unsigned int v) {
    if (LDS && sp < kLds) lstk[sp * STRIDE] = v;
    else ovf[sp - kLds] = v;
    sp++;
  };
  auto pop = [&]() -> unsigned int {
    sp--;
    if (LDS && sp < kLds) return lstk[sp * STRIDE];
    return ovf[sp - kLds];
  };
  hit_t hit;
  hit.object = -1, hit.slot = -1, hit.u = 0, hit.v = 0, hit.distance = 0;
  float tmax = ray.tmax;
  // world-space ray data for the scene level
  f3  wdinv = {1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z};
  int wsign = (wdinv.x < 0 ? 1 : 0) | (wdinv.y < 0 ? 2 : 0) | (wdinv.z < 0 ? 4 : 0);
  // instance-space ray data
  f3  lo = ray.o, ld = ray.d, ldinv = wdinv;
  int lsign = wsign, cur_obj = -1, kind = 0, node_base = 0, prim_base = 0;
  unsigned long long n_nodes = 0, n_seg = 0, n_tri = 0;
  unsigned int       n_steps = 0;

  unsigned int cur;
  if (first_object >= 0) {
    cur = YH_TAG_ENTER | (unsigned)first_object;
  } else {
    if (sc.num_scene_nodes == 0) return hit;
    cur = YH_TAG_SCENE | 0u;
  }
  while (true) {
    // ---- phase 1: nodes, until this lane holds a leaf -------------------------
    while (true) {
      if (COUNT) n_steps++;
      if (cur == YH_NONE) {
        if (sp == 0) break;
        cur = pop();
      }
      unsigned int tag = cur & YH_TAG_MASK;
      if (tag == YH_TAG_LEAF) break;
      if (tag == YH_TAG_SHAPE) {
        int idx = (int)cur;
        v4f bx0, by0, bz0, bx1, by1, bz1, rf, mt;
        int rel = idx - sc.lds_node_base;
        if (LDS && rel >= 0 && rel < sc.lds_node_count) {
          const YH_LDS v4f* n = tc.lds_nodes + 8 * rel;
          bx0 = n[0], by0 = n[1], bz0 = n[2], bx1 = n[3], by1 = n[4], bz1 = n[5], rf = n[6], mt = n[7];
        } else {
          const yhd_float4* n = sc.nodes + 8 * (size_t)idx;
          bx0 = ldg4(n), by0 = ldg4(n + 1), bz0 = ldg4(n + 2), bx1 = ldg4(n + 3), by1 = ldg4(n + 4);
          bz1 = ldg4(n + 5), rf = ldg4(n + 6), mt = ldg4(n + 7);
        }
        n_nodes++;
        bool h0 = intersect_bbox(lo, ldinv, ray.tmin, tmax, f3{bx0.x, by0.x, bz0.x}, f3{bx1.x, by1.x, bz1.x});
        bool h1 = intersect_bbox(lo, ldinv, ray.tmin, tmax, f3{bx0.y, by0.y, bz0.y}, f3{bx1.y, by1.y, bz1.y});
        bool h2 = intersect_bbox(lo, ldinv, ray.tmin, tmax, f3{bx0.z, by0.z, bz0.z}, f3{bx1.z, by1.z, bz1.z});
        bool h3 = intersect_bbox(lo, ldinv, ray.tmin, tmax, f3{bx0.w, by0.w, bz0.w}, f3{bx1.w, by1.w, bz1.w});
        unsigned int r0 = __float_as_uint(rf.x), r1 = __float_as_uint(rf.y), r2 = __float_as_uint(rf.z),
                     r3 = __float_as_uint(rf.w);
        // child wide nodes are shape-local indices
        if ((r0 & YH_TAG_MASK) == 0) r0 += (unsigned)node_base;
        if ((r1 & YH_TAG_MASK) == 0) r1 += (unsigned)node_base;
        if ((r2 & YH_TAG_MASK) == 0) r2 += (unsigned)node_base;
        if ((r3 & YH_TAG_MASK) == 0) r3 += (unsigned)node_base;
        unsigned int axes = __float_as_uint(mt.x);
        bool s0 = (lsign >> (axes & 3)) & 1, sl = (lsign >> ((axes >> 2) & 3)) & 1, sr = (lsign >> ((axes >> 4) & 3)) & 1;
        // visiting order: (left pair, right pair) or reversed by s0; inside a
        // pair (first, second) or reversed by that child's own axis sign
        unsigned int la = sl ? r1 : r0, lb = sl ? r0 : r1;  // left pair in visiting order
        bool         ha = sl ? h1 : h0, hb = sl ? h0 : h1;
        unsigned int ra = sr ? r3 : r2, rb = sr ? r2 : r3;
        bool         hc = sr ? h3 : h2, hd = sr ? h2 : h3;
        unsigned int o0 = s0 ? ra : la, o1 = s0 ? rb : lb, o2 = s0 ? la : ra, o3 = s0 ? lb : rb;
        bool         g0 = s0 ? hc : ha, g1 = s0 ? hd : hb, g2 = s0 ? ha : hc, g3 = s0 ? hb : hd;
        // push the hit children in reverse visiting order; the first one stays in `cur`
        unsigned int next = YH_NONE;
        if (g3) next = o3;
        if (g2) { if (next != YH_NONE) push(next); next = o2; }
        if (g1) { if (next != YH_NONE) push(next); next = o1; }
        if (g0) { if (next != YH_NONE) push(next); next = o0; }
        cur = next;
        continue;
      }
      if (tag == YH_TAG_ENTER) {
        // transform_ray(inverse(object.frame, true), ray) (pt.cpp:1012-1013)
        cur_obj             = (int)(cur & ~YH_TAG_MASK);
        const yhd_object& o = sc.objects[cur_obj];
        frame inv           = ldframe(o.inv_frame);
        lo                  = transform_point(inv, ray.o);
        ld                  = transform_vector(inv, ray.d);
        ldinv               = {1 / ld.x, 1 / ld.y, 1 / ld.z};
        lsign = (ldinv.x < 0 ? 1 : 0) | (ldinv.y < 0 ? 2 : 0) | (ldinv.z < 0 ? 4 : 0);
        kind = o.kind, node_base = o.node_base, prim_base = o.prim_base;
        cur = YH_TAG_SHAPE | (unsigned)node_base;  // shape root
        continue;
      }
      // scene-level node (binary, reference layout)
      int idx = (int)(cur & ~YH_TAG_MASK);
      v4f n0 = ldg4(sc.scene_nodes + 2 * idx), n1 = ldg4(sc.scene_nodes + 2 * idx + 1);
      n_nodes++;
      cur = YH_NONE;
      if (!intersect_bbox(ray.o, wdinv, ray.tmin, tmax, xyz(n0), xyz(n1))) continue;
      int start = __float_as_int(n0.w), meta = __float_as_int(n1.w);
      if (meta & 0x10000) {  // internal
        int axis = (meta >> 24) & 3;
        int near = (wsign >> axis) & 1;  // dsign set: visit start+1 first
        push(YH_TAG_SCENE | (unsigned)(start + 1 - near));
        cur = YH_TAG_SCENE | (unsigned)(start + near);
      } else {
        int num = meta & 0xffff;
        for (int i = num - 1; i >= 1; i--) push(YH_TAG_ENTER | (unsigned)sc.scene_prims[start + i]);
        if (num > 0) cur = YH_TAG_ENTER | (unsigned)sc.scene_prims[start];
      }
    }
    if (cur == YH_NONE) break;  // stack exhausted: traversal finished
    // ---- phase 2: the leaf's primitives, in leaf order (pt.cpp:905-923) -------
    int leaf_start = (int)(cur & 0x07FFFFFFu), leaf_num = (int)((cur >> 27) & 7u);
    cur = YH_NONE;
    if (kind == YH_KIND_LINES) {
      // all of the leaf's records are requested before the first test
      const yhd_float4* rec = sc.prims + (size_t)prim_base + (size_t)leaf_start * 4;
      v4f a0 = ldg4(rec), b0 = ldg4(rec + 1), a1 = a0, b1 = b0, a2 = a0, b2 = b0, a3 = a0, b3 = b0;
      if (leaf_num > 1) a1 = ldg4(rec + 4), b1 = ldg4(rec + 5);
      if (leaf_num > 2) a2 = ldg4(rec + 8), b2 = ldg4(rec + 9);
      if (leaf_num > 3) a3 = ldg4(rec + 12), b3 = ldg4(rec + 13);
      for (int i = 0; i < leaf_num; i++) {
        v4f a = i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : a3;
        v4f b = i == 0 ? b0 : i == 1 ? b1 : i == 2 ? b2 : b3;
        n_seg++;
        float uu, vv, dist;
        if (intersect_line(lo, ld, ray.tmin, tmax, xyz(a), xyz(b), a.w, b.w, uu, vv, dist)) {
          hit.object = cur_obj, hit.slot = leaf_start + i;
          hit.u = uu, hit.v = vv, hit.distance = dist;
          tmax = dist;
        }
      }
    } else {
      for (int i = 0; i < leaf_num; i++) {
        const yhd_float4* rec = sc.prims + (size_t)prim_base + (size_t)(leaf_start + i) * 6;
        v4f a = ldg4(rec), b = ldg4(rec + 1), c = ldg4(rec + 2);
        n_tri++;
        float uu, vv, dist;
        if (intersect_triangle(lo, ld, ray.tmin, tmax, xyz(a), xyz(b), xyz(c), uu, vv, dist)) {
          hit.object = cur_obj, hit.slot = leaf_start + i;
          hit.u = uu, hit.v = vv, hit.distance = dist;
          tmax = dist;
        }
      }
    }
  }
  if (COUNT) {
    if (steps_out) *steps_out = n_steps;
    count_add<COUNT>(&tc.counters->nodes, n_nodes);
    count_add<COUNT>(&tc.counters->seg, n_seg);
    count_add<COUNT>(&tc.counters->tri, n_tri);
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
