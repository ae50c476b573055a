// dev_queue.h — the traversal STAGE of the wavefront integrator (csrc/wavefront.hip).
//
// Same two-level traversal as dev_trace.h's trace_ray_loop (intersect_scene_bvh /
// intersect_shape_bvh, pt.cpp:821-1053: identical step code, identical visiting
// order, identical closest hits), but fed from a compacted ray LIST instead of
// one ray per call: a quad (four adjacent lanes, dev_trace.h) that finishes its
// ray stores the hit, classifies it (miss / hair / surface) into the lists the
// shading stage consumes, and takes the next ray of the list — so the lanes of a
// wavefront never wait for its longest ray, only for the list to run dry.
//
// Refill policy (after Aila & Laine, "Understanding the efficiency of ray
// traversal on GPUs"): the refill code is divergent with respect to the step
// code, so it runs only when at least YH_REFILL_QUADS quads of the wave are idle
// (or none is busy); one LDS atomic per refilling quad.
#ifndef YH_DEV_QUEUE_H_
#define YH_DEV_QUEUE_H_
#include "dev_trace.h"

namespace yhd {

#ifndef YH_REFILL_QUADS
#define YH_REFILL_QUADS 4
#endif

// Slot states of the path pool (one byte per slot in LDS)
enum : unsigned char { YH_SLOT_FREE = 0, YH_SLOT_RAY = 1, YH_SLOT_HAIR = 2, YH_SLOT_SURF = 3, YH_SLOT_MISS = 4, YH_SLOT_ENDED = 5 };

// LDS atomic add (ds_add_rtn_u32), workgroup scope
YH_DEV int lds_add(YH_LDS int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Appends `value` of the lanes with `pred` to an LDS list: one ballot, one LDS atomic per wave.
YH_DEV void wave_append(YH_LDS unsigned short* list, YH_LDS int* counter, bool pred, int value) {
  unsigned long long m = __ballot(pred);
  if (m == 0) return;
  const int lane   = (int)__lane_id();
  const int leader = __ffsll((long long)m) - 1;
  int       base   = 0;
  if (lane == leader) base = lds_add(counter, __popcll(m));
  base = __shfl(base, leader, 64);
  if (pred) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)value;
}

// What the traversal stage reads and writes per path slot (global memory, SoA; yh_device.h: yhd_pool).
struct queue_io {
  yhd_float4*          ray_o;     // in:  origin.xyz            out: .w = closest-hit distance
  const yhd_float4*    ray_d;     // in:  direction.xyz (w: path flags, untouched)
  yhd_int4*            hit;       // out: object (-1: miss), leaf slot, u bits, v bits
  YH_LDS unsigned char* state;    // out: YH_SLOT_HAIR / _SURF / _MISS per slot
  YH_LDS unsigned int*  work;     // += traversal trips of the ray (scheduling hint of the pixel's work item)
  YH_LDS unsigned short* hair_list;
  YH_LDS unsigned short* surf_list;
  YH_LDS unsigned short* redo_list;  // EXACT = false: rays that need the exact box test (axis-parallel)
  YH_LDS int*           n_hair;
  YH_LDS int*           n_surf;
  YH_LDS int*           n_redo;
};

// Traces the rays of slots list[0 .. count) (slot ids relative to `base`, the block's first slot).
// `head` is the LDS cursor into the list shared by the block's waves (zero on entry).
template <int STRIDE, bool EXACT>
YH_DEV void trace_queue(const trace_ctx& tc, const queue_io& io, size_t base, const YH_LDS unsigned short* list, int count,
    YH_LDS int* head) {
  const yhd_scene&     sc   = *tc.sc;
  const unsigned int   q    = __lane_id() & 3u;
  YH_LDS unsigned int* lstk = tc.lds_stack;
  auto box_test = [](f3 o, f3 dinv, float t0, float t1, f3 bmin, f3 bmax) {
    return EXACT ? intersect_bbox(o, dinv, t0, t1, bmin, bmax) : intersect_bbox_nonan(o, dinv, t0, t1, bmin, bmax);
  };
  const YH_LDS v4f* lds_snodes = tc.lds_scene ? tc.lds_scene + YH_OBJECT_F4 * sc.num_objects : nullptr;
  auto scene_prim = [&](int i) -> int {
    if (tc.lds_scene) return ((const YH_LDS int*)(lds_snodes + 2 * sc.num_scene_nodes))[i];
    return sc.scene_prims[i];
  };
  // ---- state of the ray this quad holds (identical in its four lanes) ----
  bool  have = false, dry = false;
  int   slot = 0, sp = 0;
  unsigned int cur = YH_NONE, steps = 0;
  f3    ro = mk3(0.0f), rd = mk3(0.0f), wdinv = mk3(0.0f);
  int   wsign = 0;
  bool  wnonan = true;
  float tmax = 0;
  hit_t hit;
  hit.object = -1, hit.slot = -1, hit.u = 0, hit.v = 0, hit.distance = 0;
  f3  lo = ro, ld = rd, ldinv = wdinv;
  int lsign = 0, cur_obj = -1, kind = 0, node_base = 0, prim_base = 0;
  auto push = [&](unsigned int v) { lstk[sp * STRIDE] = v, sp++; };

  while (true) {
    // ---- refill: idle quads take the next rays of the list ----
    {
      const bool want = !have && !dry;
      unsigned long long wm = __ballot(want), hm = __ballot(have);
      if (wm != 0 && (__popcll(wm) >= 4 * YH_REFILL_QUADS || hm == 0)) {
        if (want) {
          int idx = 0;
          if (q == 0) idx = lds_add(head, 1);
          idx = (int)quad_bcast_u<0>((unsigned int)idx);
          if (idx < count) {
            slot         = list[idx];
            yhd_float4 o = io.ray_o[base + slot], d = io.ray_d[base + slot];
            ro = f3{o.x, o.y, o.z}, rd = f3{d.x, d.y, d.z};
            wdinv  = quad_rcp(rd);  // one division per lane of the quad (dev_math.h)
            wsign  = (wdinv.x < 0 ? 1 : 0) | (wdinv.y < 0 ? 2 : 0) | (wdinv.z < 0 ? 4 : 0);
            wnonan = finite3(wdinv) && finite3(ro);
            tmax   = flt_max;
            hit.object = -1, hit.slot = -1, hit.u = 0, hit.v = 0, hit.distance = 0;
            lo = ro, ld = rd, ldinv = wdinv, lsign = wsign, cur_obj = -1, kind = 0, node_base = 0, prim_base = 0;
            sp = 0, steps = 0;
            cur  = sc.num_scene_nodes ? (YH_TAG_SCENE | 0u) : YH_NONE;
            have = true;
            if (!EXACT && !wnonan) {  // a slab could hold a NaN: the exact pass traces this ray
              if (q == 0) io.redo_list[lds_add(io.n_redo, 1)] = (unsigned short)slot;
              have = false;
            }
          } else {
            dry = true;
          }
        }
      }
      if (__ballot(have) == 0) {
        if (__ballot(!dry) == 0) break;  // every quad of the wave is idle and the list is empty
        continue;                        // (cannot happen: idle quads refill when none is busy)
      }
    }
    if (have) {
      bool redo = false;
      if (cur == YH_NONE && sp > 0) cur = lstk[(--sp) * STRIDE];
      steps++;
      unsigned int tag = cur & YH_TAG_MASK;
      bool         skip = cur == YH_NONE;  // only a scene without objects: the ray ends as a miss below
      if (!skip && tag == YH_TAG_SCENE) {
        int idx = (int)(cur & ~YH_TAG_MASK);
        v4f n0, n1;
        if (lds_snodes) n0 = lds_snodes[2 * idx], n1 = lds_snodes[2 * idx + 1];
        else n0 = ldg4(sc.scene_nodes + 2 * idx), n1 = ldg4(sc.scene_nodes + 2 * idx + 1);
        cur = YH_NONE;
        if (box_test(ro, wdinv, ray_eps, tmax, xyz(n0), xyz(n1))) {
          int start = __float_as_int(n0.w), meta = __float_as_int(n1.w);
          if (meta & 0x10000) {  // internal
            int axis = (meta >> 24) & 3;
            int near = (wsign >> axis) & 1;
            push(YH_TAG_SCENE | (unsigned)(start + 1 - near));
            cur = YH_TAG_SCENE | (unsigned)(start + near);
          } else {
            int num = meta & 0xffff;
            for (int i = num - 1; i >= 1; i--) push(YH_TAG_ENTER | (unsigned)scene_prim(start + i));
            if (num > 0) cur = YH_TAG_ENTER | (unsigned)scene_prim(start);
          }
        }
        tag  = cur & YH_TAG_MASK;
        skip = cur == YH_NONE || tag == YH_TAG_SCENE;
      }
      if (!skip && tag == YH_TAG_ENTER) {
        cur_obj = (int)(cur & ~YH_TAG_MASK);
        bool enter = true;
        if (wnonan) {  // padded world box of the object (dev_trace.h)
          v4f bmin, bmax;
          if (tc.lds_scene) {
            const YH_LDS v4f* ob = tc.lds_scene + YH_OBJECT_F4 * cur_obj;
            bmin = ob[8], bmax = ob[9];
          } else {
            const yhd_object& o = sc.objects[cur_obj];
            bmin = v4f{o.wbox_min[0], o.wbox_min[1], o.wbox_min[2], 0}, bmax = v4f{o.wbox_max[0], o.wbox_max[1], o.wbox_max[2], 0};
          }
          enter = box_test(ro, wdinv, ray_eps, tmax, xyz(bmin), xyz(bmax));
        }
        if (!enter) {
          cur = YH_NONE, skip = true;
        } else {
          frame inv;
          if (tc.lds_scene) {
            const YH_LDS v4f* ob = tc.lds_scene + YH_OBJECT_F4 * cur_obj;
            v4f a = ob[3], b = ob[4], c = ob[5], d = ob[6];
            inv.x = {a.x, a.y, a.z}, inv.y = {a.w, b.x, b.y}, inv.z = {b.z, b.w, c.x}, inv.o = {c.y, c.z, c.w};
            kind = __float_as_int(d.x), node_base = __float_as_int(d.y), prim_base = __float_as_int(d.z);
          } else {
            const yhd_object& o = sc.objects[cur_obj];
            inv  = ldframe(o.inv_frame);
            kind = o.kind, node_base = o.node_base, prim_base = o.prim_base;
          }
          lo    = transform_point(inv, ro);
          ld    = transform_vector(inv, rd);
          ldinv = quad_rcp(ld);
          lsign = (ldinv.x < 0 ? 1 : 0) | (ldinv.y < 0 ? 2 : 0) | (ldinv.z < 0 ? 4 : 0);
          if (!EXACT && !(finite3(ldinv) && finite3(lo))) redo = true, skip = true;
          cur = YH_TAG_SHAPE | (unsigned)node_base;
          tag = YH_TAG_SHAPE;
        }
      }
      if (!skip) {
        bool is_leaf    = tag == YH_TAG_LEAF;
        int  leaf_start = (int)(cur & 0x07FFFFFFu), leaf_num = (int)((cur >> 27) & 7u);
        int  rec        = kind == YH_KIND_LINES ? 4 : 6;
        bool mine       = !is_leaf || (int)q < leaf_num;
        int  pq         = mine ? (int)q : leaf_num - 1;
        const yhd_float4* addr = is_leaf ? sc.prims + (size_t)prim_base + (size_t)(leaf_start + pq) * rec
                                         : sc.nodes + 8 * (size_t)cur + 2 * q;
        v4f s0 = ldg4(addr), s1 = ldg4(addr + 1);
        if (!is_leaf) {
          bool h = box_test(lo, ldinv, ray_eps, tmax, f3{s0.x, s0.y, s0.z}, f3{s0.w, s1.x, s1.y});
          unsigned int ref  = __float_as_uint(s1.z);
          unsigned int axes = __float_as_uint(s1.w);
          h = h && ref != YH_NONE;
          if ((ref & YH_TAG_MASK) == 0) ref += (unsigned)node_base;
          unsigned int pair = q >> 1;
          unsigned int sgn  = (lsign >> ((axes >> (2 + 2 * pair)) & 3)) & 1;
          unsigned int s0_  = (lsign >> (axes & 3)) & 1;
          unsigned int rank = ((pair ^ s0_) << 1) | ((q & 1) ^ sgn);
          unsigned int bit  = h ? (1u << rank) : 0u;
          unsigned int M    = bit | (unsigned int)dpp_i<YH_QUAD_XOR1>((int)bit);
          M |= (unsigned int)dpp_i<YH_QUAD_XOR2>((int)M);
          bool         first = h && (M & (bit - 1)) == 0;
          unsigned int after = (unsigned int)__popc(M >> (rank + 1));
          if (h && !first) lstk[(sp + (int)after) * STRIDE] = ref;
          unsigned int mine_ref = first ? ref : 0u;
          mine_ref |= (unsigned int)dpp_i<YH_QUAD_XOR1>((int)mine_ref);
          mine_ref |= (unsigned int)dpp_i<YH_QUAD_XOR2>((int)mine_ref);
          int nh = __popc(M);
          sp += nh > 0 ? nh - 1 : 0;
          cur = nh > 0 ? mine_ref : YH_NONE;
        } else {
          cur = YH_NONE;
          bool  ok = false;
          float uu = 0, vv = 0, dist = 0;
          if (kind == YH_KIND_LINES) {
            if (mine) ok = intersect_line<true>(lo, ld, ray_eps, tmax, xyz(s0), xyz(s1), s0.w, s1.w, uu, vv, dist);
          } else {
            v4f s2 = ldg4(addr + 2);
            if (mine) ok = intersect_triangle(lo, ld, ray_eps, tmax, xyz(s0), xyz(s1), xyz(s2), uu, vv, dist);
          }
          int   key_i = ok ? (int)q : -1;
          float key_t = dist;
#define YH_QUAD_MERGE(CTRL)                                                                \
  {                                                                                        \
    int   oi = dpp_i<CTRL>(key_i);                                                         \
    float ot = dpp_f<CTRL>(key_t), ou = dpp_f<CTRL>(uu), ov = dpp_f<CTRL>(vv);             \
    bool  take = oi >= 0 && (key_i < 0 || ot < key_t || (ot == key_t && oi > key_i));      \
    if (take) key_i = oi, key_t = ot, uu = ou, vv = ov;                                    \
  }
          YH_QUAD_MERGE(YH_QUAD_XOR1)
          YH_QUAD_MERGE(YH_QUAD_XOR2)
#undef YH_QUAD_MERGE
          if (key_i >= 0) {
            hit.object = cur_obj, hit.slot = leaf_start + key_i;
            hit.u = uu, hit.v = vv, hit.distance = key_t;
            tmax = key_t;
          }
        }
      }
      // ---- ray finished (or handed to the exact pass): publish, classify ----
      const bool done = redo || (cur == YH_NONE && sp == 0);
      if (done) {
        have = false;
        if (redo) {
          if (q == 0) io.redo_list[lds_add(io.n_redo, 1)] = (unsigned short)slot;
        } else if (q == 0) {
          io.hit[base + slot] = yhd_int4{hit.object, hit.slot, __float_as_int(hit.u), __float_as_int(hit.v)};
          ((float*)&io.ray_o[base + slot])[3] = hit.distance;
          io.work[slot] += steps;
          int okind = 0;
          if (hit.object >= 0)
            okind = tc.lds_scene ? __float_as_int(tc.lds_scene[YH_OBJECT_F4 * hit.object + 6].x) : sc.objects[hit.object].kind;
          if (hit.object < 0) {
            io.state[slot] = YH_SLOT_MISS;
          } else if (okind == YH_KIND_LINES) {
            io.state[slot] = YH_SLOT_HAIR;
            io.hair_list[lds_add(io.n_hair, 1)] = (unsigned short)slot;
          } else {
            io.state[slot] = YH_SLOT_SURF;
            io.surf_list[lds_add(io.n_surf, 1)] = (unsigned short)slot;
          }
        }
      }
    }
  }
}

}  // namespace yhd
#endif
