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
  int   object, element;  // -1 on miss
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
#define YH_TAG_SHAPE 0u          /* entry = global shape-node index          */
#define YH_TAG_SCENE 0x40000000u /* entry = scene-node index                 */
#define YH_TAG_ENTER 0x80000000u /* entry = object id to enter               */
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
// The node being visited is kept in a register (`cur`): at an internal node
// the near child becomes `cur` directly and only the far child is pushed —
// the same visiting order as the reference's push(far), push(near), pop().
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
  hit.object = -1, hit.element = -1, hit.u = 0, hit.v = 0, hit.distance = 0;
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
  // "while-while" traversal: every lane first walks nodes until it owns a leaf
  // (or has nothing left), then all lanes of the wave test their leaf's
  // primitives together. A lane that reaches a leaf early waits for the others
  // instead of dragging the whole wave through the (long) primitive tests on
  // every node step.
  while (true) {
    int leaf_start = 0, leaf_num = 0;
    // ---- phase 1: nodes -----------------------------------------------------
    while (true) {
      if (COUNT) n_steps++;
      if (cur == YH_NONE) {
        if (sp == 0) break;
        cur = pop();
      }
      unsigned int tag = cur & YH_TAG_MASK;
      if (tag == YH_TAG_SHAPE) {
        int idx = (int)cur;
        v4f n0, n1;
        int rel = idx - sc.lds_node_base;
        if (LDS && rel >= 0 && rel < sc.lds_node_count) {
          n0 = tc.lds_nodes[2 * rel], n1 = tc.lds_nodes[2 * rel + 1];
        } else {
          n0 = ldg4(sc.nodes + 2 * (size_t)idx), n1 = ldg4(sc.nodes + 2 * (size_t)idx + 1);
        }
        n_nodes++;
        cur = YH_NONE;
        if (!intersect_bbox(lo, ldinv, ray.tmin, tmax, xyz(n0), xyz(n1))) continue;
        int start = __float_as_int(n0.w), meta = __float_as_int(n1.w);
        if (meta & 0x10000) {
          int axis = (meta >> 24) & 3;
          int near = (lsign >> axis) & 1;  // dsign set: visit start+1 first (pt.cpp:887-893)
          int a    = node_base + start;
          push((unsigned)(a + 1 - near));
          cur = (unsigned)(a + near);
          continue;
        }
        leaf_start = start, leaf_num = meta & 0xffff;
        break;
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
      // scene-level node
      int idx = (int)(cur & ~YH_TAG_MASK);
      v4f n0 = ldg4(sc.scene_nodes + 2 * idx), n1 = ldg4(sc.scene_nodes + 2 * idx + 1);
      n_nodes++;
      cur = YH_NONE;
      if (!intersect_bbox(ray.o, wdinv, ray.tmin, tmax, xyz(n0), xyz(n1))) continue;
      int start = __float_as_int(n0.w), meta = __float_as_int(n1.w);
      if (meta & 0x10000) {  // internal
        int axis = (meta >> 24) & 3;
        int near = (wsign >> axis) & 1;
        push(YH_TAG_SCENE | (unsigned)(start + 1 - near));
        cur = YH_TAG_SCENE | (unsigned)(start + near);
      } else {
        int num = meta & 0xffff;
        for (int i = num - 1; i >= 1; i--) push(YH_TAG_ENTER | (unsigned)sc.scene_prims[start + i]);
        if (num > 0) cur = YH_TAG_ENTER | (unsigned)sc.scene_prims[start];
      }
    }
    if (leaf_num == 0) {
      if (cur == YH_NONE && sp == 0) break;  // traversal finished
      continue;                              // empty leaf
    }
    // ---- phase 2: the leaf's primitives, in leaf order (pt.cpp:905-923) -------
    if (kind == YH_KIND_LINES) {
      for (int i = 0; i < leaf_num; i++) {
        size_t r = (size_t)prim_base + (size_t)(leaf_start + i) * 2;
        v4f    a = ldg4(sc.prims + r), b = ldg4(sc.prims + r + 1);
        n_seg++;
        float uu, vv, dist;
        if (intersect_line(lo, ld, ray.tmin, tmax, xyz(a), xyz(b), a.w, b.w, uu, vv, dist)) {
          hit.object = cur_obj, hit.element = leaf_start + i;  // leaf slot; resolved below
          hit.u = uu, hit.v = vv, hit.distance = dist;
          tmax = dist;
        }
      }
    } else {
      for (int i = 0; i < leaf_num; i++) {
        size_t r = (size_t)prim_base + (size_t)(leaf_start + i) * 3;
        v4f    a = ldg4(sc.prims + r), b = ldg4(sc.prims + r + 1), c = ldg4(sc.prims + r + 2);
        n_tri++;
        float uu, vv, dist;
        if (intersect_triangle(lo, ld, ray.tmin, tmax, xyz(a), xyz(b), xyz(c), uu, vv, dist)) {
          hit.object = cur_obj, hit.element = ~__float_as_int(a.w);  // already an element id
          hit.u = uu, hit.v = vv, hit.distance = dist;
          tmax = dist;
        }
      }
    }
  }
  // hair hits carry the leaf slot: one dependent load of the element id, only
  // for the final closest hit (triangle hits were stored complemented)
  if (hit.object >= 0) {
    if (hit.element >= 0) {
      const yhd_object& o = sc.objects[hit.object];
      hit.element         = sc.prim_elem[o.slot_base + hit.element];
    } else {
      hit.element = ~hit.element;
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

}  // namespace yhd
#endif
