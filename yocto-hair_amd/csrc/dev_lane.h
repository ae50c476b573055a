// dev_lane.h — the two-level BVH traversal with ONE LANE PER RAY (csrc/stream.hip).
//
// Same algorithm, same visiting order and the same closest hits as dev_trace.h's quad form
// (intersect_scene_bvh / intersect_shape_bvh, pt.cpp:821-1053) — what changes is who does the work:
//
//   dev_trace.h   a QUAD of four lanes per ray: one box / one primitive per lane, everything else
//                 repeated four times. Right when rays are scarce (a few expensive pixels bound the
//                 launch: the chain of one path is what counts).
//   dev_lane.h    one lane per ray, 64 rays per wavefront: a node step tests the four slots of the
//                 wide node one after the other; a line leaf's segments are tested by the WAVE, one
//                 per lane, and the leaf's lane applies the reference's sequential accept rule to
//                 the results (lane_step, COOP); a triangle leaf (and every leaf of the out-of-line
//                 exact form) is tested by its own lane, two primitives per step, tmax shrinking
//                 after each accepted hit (pt.cpp:905-923). Nothing is computed twice, so a ray
//                 costs about a third of the vector instructions; right when every pixel is
//                 expensive (dense hair) and throughput is what counts.
//
// A step is still ONE dependent round trip: a lane loads the four 32-byte slots of its node (or the
// test halves of its triangle leaf) and the one segment it tests for the wave, back to back.
//
// The traversal is a STEP function over explicit state (lane_trav), not a loop, so that the caller
// can refill finished lanes from a ray list between steps and SUSPEND the unfinished rays of a wave
// while it shades (stream.hip): a ray's state survives in registers, its stack in the lane's column.
//
// Stack: a window of YH_LSTACK entries per lane in LDS (column `lane` of a [depth][64] array,
// conflict-free) over an overflow array in global memory ([depth][64] per wave, coalesced). The
// window holds the top of the stack; pushing into a full window spills its oldest entry, popping
// below it reads the overflow array. Typical depths never leave the window.
#ifndef YH_DEV_LANE_H_
#define YH_DEV_LANE_H_
#include "dev_trace.h"

namespace yhd {

#define YH_LSTACK 16 /* LDS stack window per lane, entries (power of two) */
// Developer instrumentation of a step, both compiled out of the product kernels:
//   YH_ISA_MARKS   comment lines in the assembly at the start of every branch of lane_step (tools/isa_blocks.py --marks
//                  counts the instructions between them: the static half of profiles/r04/k_stream_branch_budget.txt)
//   PROF           (template argument) per branch: how many wave steps ran it, with how many lanes (the dynamic half).
//                  `pc` is per lane: a lane in a branch taken by n lanes adds 1 / n and 1, so the sums over the wave's
//                  lanes (the caller's job) are the executions of the branch and the lanes in them
#ifdef YH_ISA_MARKS
#define YH_MARK(name) asm volatile("; YHMARK " name)
#else
#define YH_MARK(name)
#endif
enum { LP_STEP = 0, LP_POP, LP_SCENE, LP_ENTER, LP_FETCH, LP_NODE, LP_LINE_LEAF, LP_TRI_LEAF, LP_PUSH, LP_SEGS, LP_COUNT };  // pc[2 b]: wave steps that ran branch b, pc[2 b + 1]: lanes in them (summed over the lanes)
#define YH_LPROF(b)                                                                   \
  if (PROF && pc) {                                                                   \
    pc[2 * (b)] += 1.0f / (float)__popcll(__ballot(true)), pc[2 * (b) + 1] += 1.0f; \
  }

struct lane_stack {
  YH_LDS unsigned int* lds;  // this lane's column of the window: entry i at lds[(i & (YH_LSTACK - 1)) * 64]
  unsigned int*        ovf;  // this lane's column of the overflow array: entry i at ovf[i * 64]
  int                  sp;   // entries on the stack
  int                  base; // entries [base, sp) are in the window, [0, base) in the overflow array
};
YH_DEV void lane_push(lane_stack& s, unsigned int v) {
  if (s.sp - s.base == YH_LSTACK) {  // window full: its oldest entry goes to memory
    s.ovf[(size_t)s.base * 64] = s.lds[(s.base & (YH_LSTACK - 1)) * 64];
    s.base++;
  }
  s.lds[(s.sp & (YH_LSTACK - 1)) * 64] = v;
  s.sp++;
}
YH_DEV unsigned int lane_pop(lane_stack& s) {
  s.sp--;
  if (s.sp < s.base) {  // below the window
    s.base = s.sp;
    return s.ovf[(size_t)s.sp * 64];
  }
  return s.lds[(s.sp & (YH_LSTACK - 1)) * 64];
}

// State of one ray in flight.
struct lane_trav {
  f3           ro, rd, wdinv;  // world-space ray, 1 / d
  f3           lo, ld, ldinv;  // the ray in the space of the object it is in
  int          wsign, lsign;   // sign bits of 1 / d (x | y << 1 | z << 2)
  int          cur_obj, kind;
  unsigned int cur;            // the entry being visited (YH_NONE: pop the next one)
  float        tmax;
  // While the ray is in flight the hit is kept RAW — hit.slot = the primitive's test record (32-byte units of
  // lane_blob) and, on a line, hit.v = the squared distance d2 with hit_r = the radius there: the reference's
  // uv.y = sqrt(d2) / r (math.h:3465) is evaluated once, for the ray's final hit (lane_hit), not at every accepted test
  hit_t        hit;
  float        hit_r;
  float        ld2;            // dot(ld, ld): the `a` of every line test of this ray in this object (math.h:3437)
  bool         hit_lines;      // the closest hit so far is on a line shape (hair)
  unsigned int steps;
  bool         wnonan;         // no slab of a world-space box test can hold a NaN (dev_trace.h)
};

// Starts a ray: the whole scene (first_object < 0) or one instance (intersect_instance_bvh, pt.cpp:1031-1037).
YH_DEV void lane_begin(const yhd_scene& sc, lane_trav& t, f3 ro, f3 rd, int first_object) {
  t.ro = ro, t.rd = rd;
  t.wdinv  = f3{1 / rd.x, 1 / rd.y, 1 / rd.z};
  t.wsign  = (t.wdinv.x < 0 ? 1 : 0) | (t.wdinv.y < 0 ? 2 : 0) | (t.wdinv.z < 0 ? 4 : 0);
  t.wnonan = finite3(t.wdinv) && finite3(ro);
  t.lo = ro, t.ld = rd, t.ldinv = t.wdinv, t.lsign = t.wsign;
  t.cur_obj = -1, t.kind = 0;
  t.tmax = flt_max, t.steps = 0;
  t.hit.object = -1, t.hit.slot = -1, t.hit.u = 0, t.hit.v = 0, t.hit.distance = 0;
  t.hit_lines = false, t.hit_r = 1.0f, t.ld2 = dot(rd, rd);
  if (first_object >= 0) t.cur = YH_TAG_ENTER | (unsigned)first_object;
  else t.cur = sc.num_scene_nodes ? (YH_TAG_SCENE | 0u) : YH_NONE;
}

// The line test of math.h:3426-3469 in dev_trace.h's STRAIGHT form, with `a` = dot(rd, rd) handed in (the same for every
// segment a ray meets inside one object) and WITHOUT the uv.y = sqrt(d2) / r of an accepted hit: s, d2 and r are returned and
// the caller evaluates it for the hit that survives (same operands, same operations: same bits).
YH_DEV bool intersect_line_raw(f3 ro, f3 rd, float a, float tmin, float tmax, f3 p0, f3 p1, float r0, float r1, float& s_out, float& d2_out,
    float& r_out, float& dist) {
  f3    v = p1 - p0, w = ro - p0;
  float b = dot(rd, v), c = dot(v, v), d = dot(rd, w), e = dot(v, w);
  float det = a * c - b * b;
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
  s_out = s, d2_out = d2, r_out = r, dist = t;
  return ok;
}

// The finished ray's hit as the rest of the code knows it (hit_t of dev_trace.h: slot = leaf-order index of the primitive
// in its shape, uv as the reference's intersect_line / intersect_triangle return them) from the raw form above.
YH_DEV hit_t lane_hit(const trace_ctx& tc, hit_t raw, bool hit_lines, float hit_r) {
  if (raw.object >= 0) {
    int lane_test;
    if (tc.lds_scene) lane_test = __float_as_int(tc.lds_scene[YH_OBJECT_F4 * raw.object + 10].y);
    else lane_test = tc.sc->objects[raw.object].lane_test;
    raw.slot = hit_lines ? raw.slot - lane_test : (raw.slot - lane_test) >> 1;
    if (hit_lines) raw.v = sqrtf(raw.v) / hit_r;
  }
  return raw;
}

// ... and when the traversal kept NOTHING of a line hit but its place and distance (the cooperative leaves of lane_step: the segment was tested
// in another lane, and u, d2, r are not carried back per step): the final hit's uv from the test itself, once per ray — the ray into the
// object's space as ENTER takes it there, the segment's record, the same arithmetic (intersect_line_raw): same operands, same bits.
YH_DEV hit_t lane_hit_retest(const trace_ctx& tc, hit_t raw, bool hit_lines, f3 ro, f3 rd) {
  if (raw.object >= 0 && hit_lines) {
    frame inv;
    int   prim_base, lane_test;
    if (tc.lds_scene) {
      const YH_LDS v4f* ob = tc.lds_scene + YH_OBJECT_F4 * raw.object;
      v4f a = ob[3], b = ob[4], c = ob[5];
      inv.x = {a.x, a.y, a.z}, inv.y = {a.w, b.x, b.y}, inv.z = {b.z, b.w, c.x}, inv.o = {c.y, c.z, c.w};
      prim_base = __float_as_int(ob[6].z), lane_test = __float_as_int(ob[10].y);
    } else {
      const yhd_object& o = tc.sc->objects[raw.object];
      inv = ldframe(o.inv_frame), prim_base = o.prim_base, lane_test = o.lane_test;
    }
    const f3          lo = transform_point(inv, ro), ld = transform_vector(inv, rd);
    // the segment's {p0, r0}{p1, r1} from its LEAF RECORD (yhd_scene::prims: the blob's test record is a copy of these 32 bytes): the shading stage that
    // follows reads the same record for the tangents (eval_hit), so a shaded hair hit costs one cache line, not one of the blob's and one of the records'
    const yhd_float4* a  = tc.sc->prims + (size_t)prim_base + 4 * (size_t)(raw.slot - lane_test);
    const v4f         A = ldg4(a), B = ldg4(a + 1);
    float ss, d2, rr, dist;
    (void)intersect_line_raw(lo, ld, dot(ld, ld), ray_eps, flt_max, xyz(A), xyz(B), A.w, B.w, ss, d2, rr, dist);
    raw.u = ss, raw.v = d2;
    return lane_hit(tc, raw, true, rr);
  }
  return lane_hit(tc, raw, hit_lines, 1.0f);
}

// One step of the ray in `t`. Returns true when the ray is finished (closest hit in t.hit), or —
// EXACT = false only — when it has to be traced again by the EXACT form (`redo` set: a slab of a
// box test could hold a NaN, dev_trace.h). `sp0` = stack height at which this ray started.
//
// COOP (k_stream, k_intersect_lanes: every lane of the wave calls the step together, `active` = this lane holds a ray, `cmap` = 64 eight-byte
// entries of the wave's own LDS): LINE LEAVES ARE TESTED BY THE WAVE, not by the lane that reached them. A leaf of the reference's tree holds
// 1-4 segments (2.9 on average) and 12-17 of a wave's 64 lanes are at one in a step; tested by their own lanes, two segments one after the
// other, the line test — the largest block of the step, 205 of 350 vector instructions — ran for a quarter of the lanes and a leaf took 1.67
// steps. Here the segments of all those leaves are DEALT one per lane over the wave: a leaf lane's first test goes to lane `start` = the
// number of tests of the leaf lanes below it (three ballots of the count's bits), it writes {itself, which segment, the record's place} for
// each into cmap[start + i], lane j reads cmap[j], loads that one 32-byte record in the step's one round trip, pulls the ray of its source
// lane across the wave (ds_bpermute: origin, direction, a, tmax) and runs the test ONCE per step, with ~ 35 lanes busy instead of 16. The
// source lane pulls the distances of its tests back, applies the reference's sequential accept rule in leaf order (pt.cpp:905-923: a test
// is accepted when !(t > tmax) with tmax = the last accepted distance — minimum t, the later segment on a tie; a test made with the step's
// first tmax and re-checked against the running one gives the same answer) and pulls u, d2, r of the survivor. Tests that do not fit the 64
// lanes stay in the entry for the next step. Triangle leaves and nodes are the lane's own as before.
template <bool EXACT, bool PROF = false, bool COOP = false>
YH_DEV bool lane_step(const trace_ctx& tc, lane_trav& t, lane_stack& s, int sp0, bool& redo, float* pc = nullptr, bool active = true,
    YH_LDS unsigned long long* cmap = nullptr, unsigned long long* tacc = nullptr) {
  const yhd_scene& sc = *tc.sc;
  // PROF + COOP: where a step's time goes — shader-clock stamps at five points of the step, summed per wave in tacc[0..4] (head | exchange and
  // loads issued | the wait for memory + node code | the wave's line tests | results back and accept); csrc/stream.hip adds them to its counters
  unsigned long long tp = 0;
  auto stamp = [&](int k) {
    if (PROF && COOP && tacc) {
      const unsigned long long now = (unsigned long long)clock64();
      if (k >= 0) tacc[k] += now - tp;
      tp = now;
    }
  };
  stamp(-1);
  YH_MARK("step_begin");
  if (!COOP || active) { YH_LPROF(LP_STEP) }
  auto box_test = [](f3 o, f3 dinv, float t0, float t1, f3 bmin, f3 bmax) {
    return EXACT ? intersect_bbox(o, dinv, t0, t1, bmin, bmax) : intersect_bbox_nonan(o, dinv, t0, t1, bmin, bmax);
  };
  const YH_LDS v4f* lds_snodes = tc.lds_scene ? tc.lds_scene + YH_OBJECT_F4 * sc.num_objects : nullptr;
  auto scene_prim = [&](int i) -> int {
    if (tc.lds_scene) return ((const YH_LDS int*)(lds_snodes + 2 * sc.num_scene_nodes))[i];
    return sc.scene_prims[i];
  };
  unsigned int tag  = 0;
  bool         skip = true;
  // The head of a step — pop, scene level, ENTER: what the lane does before its fetch. false: the ray has to be traced again (redo).
  auto head = [&]() -> bool {
  if (!EXACT && !t.wnonan) {
    redo = true;
    return false;
  }
  if (t.cur == YH_NONE && s.sp > sp0) {
    YH_MARK("pop");
    YH_LPROF(LP_POP)
    t.cur = lane_pop(s);
  }
  YH_MARK("step_head");
  t.steps++;
  tag  = t.cur & YH_TAG_MASK;
  skip = t.cur == YH_NONE;  // only a scene without objects
  if (!skip && tag == YH_TAG_SCENE) {  // scene-level node (binary, the reference's layout)
    YH_MARK("scene");
    YH_LPROF(LP_SCENE)
    int idx = (int)(t.cur & ~YH_TAG_MASK);
    v4f n0, n1;
    if (lds_snodes) n0 = lds_snodes[2 * idx], n1 = lds_snodes[2 * idx + 1];
    else n0 = ldg4(sc.scene_nodes + 2 * idx), n1 = ldg4(sc.scene_nodes + 2 * idx + 1);
    t.cur = YH_NONE;
    if (box_test(t.ro, t.wdinv, ray_eps, t.tmax, xyz(n0), xyz(n1))) {
      int start = __float_as_int(n0.w), meta = __float_as_int(n1.w);
      if (meta & 0x10000) {  // internal: near side first (pt.cpp:995-1003)
        int axis = (meta >> 24) & 3;
        int near = (t.wsign >> axis) & 1;
        lane_push(s, YH_TAG_SCENE | (unsigned)(start + 1 - near));
        t.cur = YH_TAG_SCENE | (unsigned)(start + near);
      } else {
        int num = meta & 0xffff;
        for (int i = num - 1; i >= 1; i--) lane_push(s, YH_TAG_ENTER | (unsigned)scene_prim(start + i));
        if (num > 0) t.cur = YH_TAG_ENTER | (unsigned)scene_prim(start);
      }
    }
    tag  = t.cur & YH_TAG_MASK;
    skip = t.cur == YH_NONE || tag == YH_TAG_SCENE;
  }
  if (!skip && tag == YH_TAG_ENTER) {  // transform_ray(inverse(object.frame, true), ray) (pt.cpp:1012-1013)
    YH_MARK("enter");
    YH_LPROF(LP_ENTER)
    t.cur_obj  = (int)(t.cur & ~YH_TAG_MASK);
    bool enter = true;
    if (t.wnonan) {  // the object's padded world box (dev_trace.h)
      v4f bmin, bmax;
      if (tc.lds_scene) {
        const YH_LDS v4f* ob = tc.lds_scene + YH_OBJECT_F4 * t.cur_obj;
        bmin = ob[8], bmax = ob[9];
      } else {
        const yhd_object& o = sc.objects[t.cur_obj];
        bmin = v4f{o.wbox_min[0], o.wbox_min[1], o.wbox_min[2], 0}, bmax = v4f{o.wbox_max[0], o.wbox_max[1], o.wbox_max[2], 0};
      }
      enter = box_test(t.ro, t.wdinv, ray_eps, t.tmax, xyz(bmin), xyz(bmax));
    }
    if (!enter) {
      t.cur = YH_NONE, skip = true;
    } else {
      frame inv;
      if (tc.lds_scene) {
        const YH_LDS v4f* ob = tc.lds_scene + YH_OBJECT_F4 * t.cur_obj;
        v4f a = ob[3], b = ob[4], c = ob[5], d = ob[6];
        inv.x = {a.x, a.y, a.z}, inv.y = {a.w, b.x, b.y}, inv.z = {b.z, b.w, c.x}, inv.o = {c.y, c.z, c.w};
        t.kind = __float_as_int(d.x);
      } else {
        const yhd_object& o = sc.objects[t.cur_obj];
        inv    = ldframe(o.inv_frame);
        t.kind = o.kind;
      }
      t.lo    = transform_point(inv, t.ro);
      t.ld    = transform_vector(inv, t.rd);
      t.ldinv = f3{1 / t.ld.x, 1 / t.ld.y, 1 / t.ld.z};
      t.lsign = (t.ldinv.x < 0 ? 1 : 0) | (t.ldinv.y < 0 ? 2 : 0) | (t.ldinv.z < 0 ? 4 : 0);
      if (!EXACT && !(finite3(t.ldinv) && finite3(t.lo))) {
        redo = true;
        return false;
      }
      {  // the shape's root in the blob (yhd_object::lane_root): fetched in this same step
        int root;
        if (tc.lds_scene) root = __float_as_int(tc.lds_scene[YH_OBJECT_F4 * t.cur_obj + 10].x);
        else root = sc.objects[t.cur_obj].lane_root;
        t.cur = (unsigned)root;
        t.ld2 = dot(t.ld, t.ld);
      }
      tag   = YH_TAG_SHAPE;
    }
  }
  return true;
  };
  bool aborted = false;
  if (COOP) {
    if (active) aborted = !head();
    if (aborted) skip = true;
  } else {
    if (!active) return false;
    if (!head()) return true;
  }
  YH_MARK("after_enter");
  v4f A0, B0, A1, B1, A2, B2, A3, B3;
  // ---- wide node: the four slots {min.xyz, max.x} {max.yz, ref, axes}; refs are blob offsets, bits 8-11 of axes = occupied slots ----
  auto node_code = [&]() {
      YH_MARK("node");
      YH_LPROF(LP_NODE)
      const unsigned int axes = __float_as_uint(B0.w);
      const unsigned int r0 = __float_as_uint(B0.z), r1 = __float_as_uint(B1.z), r2 = __float_as_uint(B2.z), r3 = __float_as_uint(B3.z);
      if (COOP) asm volatile("" ::"v"(B1.w), "v"(B2.w), "v"(B3.w));  // (words nobody reads: their registers stay the loads' until the loads are back — handed to another value, that value's write would wait for the load)
      unsigned int hm = 0;
      hm |= box_test(t.lo, t.ldinv, ray_eps, t.tmax, f3{A0.x, A0.y, A0.z}, f3{A0.w, B0.x, B0.y}) ? 1u : 0u;
      hm |= box_test(t.lo, t.ldinv, ray_eps, t.tmax, f3{A1.x, A1.y, A1.z}, f3{A1.w, B1.x, B1.y}) ? 2u : 0u;
      hm |= box_test(t.lo, t.ldinv, ray_eps, t.tmax, f3{A2.x, A2.y, A2.z}, f3{A2.w, B2.x, B2.y}) ? 4u : 0u;
      hm |= box_test(t.lo, t.ldinv, ray_eps, t.tmax, f3{A3.x, A3.y, A3.z}, f3{A3.w, B3.x, B3.y}) ? 8u : 0u;
      hm &= axes >> 8;
      // Visiting order of the slots (pt.cpp:887-893 at both collapsed levels, dev_trace.h): the pair on the near side of
      // the node's axis first, inside a pair the slot on the near side of that child's axis. The first hit slot in that order
      // becomes `cur`, the others go on the stack so that they pop in order: the k-th visited hit (k >= 1) lands n - 1 - k
      // entries above the old top, n = the hits. NO BRANCH per slot: the rank of a hit among the hits visited before it
      // places it, three predicated stores do the pushes (round 5: the loop that pushed one entry per hit behind two nested
      // branches ran all of its three bodies in nearly every step, some forty vector and thirty scalar instructions).
      const unsigned int s0  = ((unsigned)t.lsign >> (axes & 3)) & 1;
      const unsigned int sg0 = ((unsigned)t.lsign >> ((axes >> 2) & 3)) & 1, sg1 = ((unsigned)t.lsign >> ((axes >> 4) & 3)) & 1;
      YH_MARK("node_order");
      unsigned int vq[4], vh[4], vref[4];  // by visiting order r: the slot, whether it is hit, its reference
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const unsigned int pair = ((unsigned)r >> 1) ^ s0;
        vq[r]   = (pair << 1) | (((unsigned)r & 1) ^ (pair ? sg1 : sg0));
        vh[r]   = (hm >> vq[r]) & 1u;
        vref[r] = (vq[r] & 2) ? ((vq[r] & 1) ? r3 : r2) : ((vq[r] & 1) ? r1 : r0);
      }
      const int np = max((int)(vh[0] + vh[1] + vh[2] + vh[3]) - 1, 0);  // entries to push
      if (np > 0) { YH_LPROF(LP_PUSH) }
      while (s.sp - s.base + np > YH_LSTACK) {  // (rare: the window's oldest entries go to memory, lane_push's rule)
        s.ovf[(size_t)s.base * 64] = s.lds[(s.base & (YH_LSTACK - 1)) * 64];
        s.base++;
      }
      t.cur = vh[0] ? vref[0] : vh[1] ? vref[1] : vh[2] ? vref[2] : vh[3] ? vref[3] : YH_NONE;
      unsigned int before = vh[0];  // hits visited before slot r
#pragma unroll
      for (int r = 1; r < 4; r++) {
        if (vh[r] && before >= 1) s.lds[((s.sp + np - (int)before) & (YH_LSTACK - 1)) * 64] = vref[r];
        before += vh[r];
      }
      s.sp += np;
  };
  // ---- triangle leaf: two triangles per step in leaf order, tmax shrinking after each accepted hit (pt.cpp:905-923) ----
  auto tri_code = [&](unsigned int off, int leaf_num) {
        YH_MARK("tri_leaf");
        YH_LPROF(LP_TRI_LEAF)
#define YH_LANE_ACCEPT_TRI(I)                                       \
  if (ok && I < leaf_num) {                                         \
    t.hit.object = t.cur_obj, t.hit.slot = (int)off + 2 * I;        \
    t.hit.u = uu, t.hit.v = vv, t.hit.distance = dist;              \
    t.tmax = dist, t.hit_lines = false;                             \
  }
        {
          float uu = 0, vv = 0, dist = 0;
          bool  ok = intersect_triangle(t.lo, t.ld, ray_eps, t.tmax, xyz(A0), xyz(B0), xyz(A1), uu, vv, dist);
          YH_LANE_ACCEPT_TRI(0)
        }
        if (leaf_num > 1) {
          float uu = 0, vv = 0, dist = 0;
          bool  ok = intersect_triangle(t.lo, t.ld, ray_eps, t.tmax, xyz(A2), xyz(B2), xyz(A3), uu, vv, dist);
          YH_LANE_ACCEPT_TRI(1)
        }
#undef YH_LANE_ACCEPT_TRI
  };
  if (COOP) {
    const int          lane     = (int)__lane_id();
    const bool         is_leaf  = !skip && tag == YH_TAG_LEAF;
    const bool         lines    = t.kind == YH_KIND_LINES;
    const bool         lf       = is_leaf && lines;
    const int          leaf_num = (int)((t.cur >> 27) & 7u);
    const unsigned int off      = t.cur & (is_leaf ? 0x07FFFFFFu : 0x3FFFFFFFu);  // 32-byte units into the blob
    stamp(0);
    YH_MARK("fetch");
    // ONE round trip for the lane's own node (or triangle leaf) and for the segment it tests for the wave, through BUFFER LOADS: a lane that
    // has nothing to fetch hands the load an offset beyond the end of the array — the address unit answers zeros without touching memory — so
    // every load is unconditional. No branch around any of them: the compiler counts what is in flight exactly (with loads in two arms of a
    // branch it waits for ALL of them at the join, and the first form of this step made two dependent round trips that way:
    // profiles/r05/coop_line_leaves.txt), the lane's own node goes out BEFORE the exchange that finds the segments' lanes (its LDS round trip
    // runs under the memory's), and a lane without a test costs the segment loads nothing. (32-bit offsets: the array has to end below 4 GB,
    // host/launch_plan.cpp: lane_kernels_can_address.)
    typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
    const unsigned int blob_bytes = (unsigned int)min((long long)sc.lane_blob_units * 32ll, 0xFFFFFE00ll);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sc.lane_blob, 0, (int)blob_bytes, 0x00020000);
    auto bload = [&](unsigned int byte_off) { v4u_ r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0); return v4f{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)}; };
    const bool own = !skip && !lf;
    {
      const unsigned int o = own ? off * 32u : 0xFFFFFF00u;
      if (own) { YH_LPROF(LP_FETCH) }
      A0 = bload(o), B0 = bload(o + 16u), A1 = bload(o + 32u), B1 = bload(o + 48u), A2 = bload(o + 64u), B2 = bload(o + 80u), A3 = bload(o + 96u), B3 = bload(o + 112u);
    }
    // the tests of the wave's line leaves, dealt over its lanes
    const int                c  = lf ? min(leaf_num, 4) : 0;
    const unsigned long long m0 = __ballot((c & 1) != 0), m1 = __ballot((c & 2) != 0), m2 = __ballot((c & 4) != 0);
    const bool               any_leaf = (m0 | m1 | m2) != 0;  // (wave-uniform)
    auto below = [](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
    int       start = 0, g = 0, total = 0;
    bool      work  = false;
    unsigned int w_off = 0, w_src = 0;
    if (any_leaf) {
      start = below(m0) + 2 * below(m1) + 4 * below(m2);
      total = (int)__popcll(m0) + 2 * (int)__popcll(m1) + 4 * (int)__popcll(m2);
      g     = max(0, min(c, 64 - start));
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (i < g) cmap[start + i] = ((unsigned long long)(off + (unsigned)i) << 32) | (unsigned long long)(unsigned)lane;  // {the record's place, this lane}
      work = lane < total;
      if (work) {
        const unsigned long long e = cmap[lane];
        w_off = (unsigned int)(e >> 32), w_src = (unsigned int)e;
      }
    }
    const unsigned int so = work ? w_off * 32u : 0xFFFFFF00u;
    const v4f S0 = bload(so), S1 = bload(so + 16u);
    stamp(1);
    if (own && !is_leaf) node_code();
    if (own && is_leaf) {
      const unsigned int cur_next = leaf_num > 2 ? (YH_TAG_LEAF | ((unsigned)(leaf_num - 2) << 27) | (off + 4u)) : YH_NONE;
      tri_code(off, leaf_num);
      t.cur = cur_next;
    }
    // the source lane's ray, across the wave (every lane takes part in the exchange: a pull reads the registers of a lane that may itself be
    // idle). Behind the node code, not in front of it: eight more values alive across it put spill reloads into the test, and an LDS round trip
    // is not what a step waits for (profiles/r05/coop_line_leaves.txt)
    auto pull = [](int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); };
    f3    wlo = mk3(0.0f), wld = mk3(0.0f);
    float wa = 0.0f, wtmax = 0.0f;
    if (any_leaf) {
      const int sa = (int)(w_src << 2);
      wlo = f3{pull(sa, t.lo.x), pull(sa, t.lo.y), pull(sa, t.lo.z)}, wld = f3{pull(sa, t.ld.x), pull(sa, t.ld.y), pull(sa, t.ld.z)};
      wa = pull(sa, t.ld2), wtmax = pull(sa, t.tmax);
    }
    stamp(2);
    if (any_leaf) {
      YH_MARK("line_leaf");
      float key = -1.0f;  // key: the distance of an accepted test, -1 otherwise (an accepted t is >= ray_eps — or a NaN, which the rule below lets through as the reference's comparisons do)
      if (work) {
        YH_LPROF(LP_LINE_LEAF)
        float ss, d2, rr, dist;
        const bool ok = intersect_line_raw(wlo, wld, wa, ray_eps, wtmax, xyz(S0), xyz(S1), S0.w, S1.w, ss, d2, rr, dist);
        key = ok ? dist : -1.0f;
      }
      stamp(3);
      // back to the leaf's lane: the reference's accept rule over its tests in leaf order
      float best = t.tmax;
      int   win  = -1;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float ki  = pull((start + i) << 2, key);  // (start + i < 64 wherever i < g; the others' answers are not looked at)
        const bool  acc = (i < g) & !(ki < 0.0f) & !(ki > best);
        best = acc ? ki : best, win = acc ? i : win;
      }
      if (lf) {
        if (win >= 0) {  // (place and distance only: u, d2, r of the ray's FINAL hit come from lane_hit_retest, once per ray)
          t.hit.object = t.cur_obj, t.hit.slot = (int)off + win;
          t.hit.distance = best, t.tmax = best, t.hit_lines = true;
        }
        const int rest = leaf_num - g;
        t.cur = rest > 0 ? (YH_TAG_LEAF | ((unsigned)rest << 27) | (off + (unsigned)g)) : YH_NONE;
      }
    }
    stamp(4);
    YH_MARK("step_end");
    return active && (aborted || (t.cur == YH_NONE && s.sp == sp0));
  }
  if (!skip) {
    YH_MARK("fetch");
    YH_LPROF(LP_FETCH)
    const bool         is_leaf  = tag == YH_TAG_LEAF;
    const bool         lines    = t.kind == YH_KIND_LINES;
    const int          leaf_num = (int)((t.cur >> 27) & 7u);
    const unsigned int off      = t.cur & (is_leaf ? 0x07FFFFFFu : 0x3FFFFFFFu);  // 32-byte units into the blob
    // Whatever the lane holds, its record is at lane_blob + 32 * off:
    //   wide node      slot q = {A_q, B_q}
    //   line leaf      segment i = {p0 r0, p1 r1} = {A_i, B_i}   (two per step)
    //   triangle leaf  triangle i = {p0}{p1}{p2}{-} = {A_2i, B_2i, A_2i+1}   (two per step)
    // ONE round trip, one 64-bit address; the second 64 bytes are fetched only by the lanes that use them.
    const yhd_float4* a = sc.lane_blob + 2 * (size_t)off;
    A0 = ldg4(a), B0 = ldg4(a + 1), A1 = ldg4(a + 2), B1 = ldg4(a + 3);
    if (!is_leaf || (!lines && leaf_num > 1)) A2 = ldg4(a + 4), B2 = ldg4(a + 5), A3 = ldg4(a + 6), B3 = ldg4(a + 7);
    if (!is_leaf) {
      node_code();
    } else {
      YH_MARK("leaf");
      // ---- leaf: its primitives in leaf order, tmax shrinking after each accepted hit (pt.cpp:905-923) ----
      t.cur = leaf_num > 2 ? (YH_TAG_LEAF | ((unsigned)(leaf_num - 2) << 27) | (off + (lines ? 2u : 4u))) : YH_NONE;  // two primitives per step
#define YH_LANE_ACCEPT_LINE(I)                                          \
  if (ok && I < leaf_num) {                                             \
    t.hit.object = t.cur_obj, t.hit.slot = (int)off + I;                \
    t.hit.u = ss, t.hit.v = d2, t.hit_r = rr, t.hit.distance = dist;    \
    t.tmax = dist, t.hit_lines = true;                                  \
  }
      if (lines) {
        YH_MARK("line_leaf");
        YH_LPROF(LP_LINE_LEAF)
        if (PROF && leaf_num > 1) { YH_LPROF(LP_SEGS) }  // lanes whose second test of the step is a real segment
        {
          float ss, d2, rr, dist;
          bool  ok = intersect_line_raw(t.lo, t.ld, t.ld2, ray_eps, t.tmax, xyz(A0), xyz(B0), A0.w, B0.w, ss, d2, rr, dist);
          YH_LANE_ACCEPT_LINE(0)
        }
        {
          float ss, d2, rr, dist;
          bool  ok = intersect_line_raw(t.lo, t.ld, t.ld2, ray_eps, t.tmax, xyz(A1), xyz(B1), A1.w, B1.w, ss, d2, rr, dist);
          YH_LANE_ACCEPT_LINE(1)
        }
      } else {
        tri_code(off, leaf_num);
      }
#undef YH_LANE_ACCEPT_LINE
    }
  }
  YH_MARK("step_end");
  return t.cur == YH_NONE && s.sp == sp0;
}

// A whole ray with the reference's compare-and-select box test throughout (axis-parallel rays: rare).
// Out of line — one copy of the EXACT step for every caller — and everything by value, so that the
// callers' stack and ray state stay in registers.
struct lane_exact_result {
  hit_t hit;    // raw, as lane_trav keeps it (lane_hit gives the final form)
  float hit_r;
  int   hit_lines;
  int   base;  // the stack's window base afterwards (sp is back where it was)
};
__device__ __attribute__((noinline)) lane_exact_result lane_trace_exact(const yhd_scene* sc, const YH_LDS v4f* lds_scene,
    YH_LDS unsigned int* lds, unsigned int* ovf, int sp, int base, f3 ro, f3 rd, int first_object) {
  trace_ctx tc;
  tc.sc = sc, tc.lds_stack = nullptr, tc.lds_scene = lds_scene, tc.stats = nullptr, tc.ls = nullptr, tc.sc_dev = sc;
  tc.lds_lights = nullptr, tc.lds_envtab = nullptr, tc.lds_mats = nullptr;
  lane_stack s;
  s.lds = lds, s.ovf = ovf, s.sp = sp, s.base = base;
  lane_trav t;
  lane_begin(*sc, t, ro, rd, first_object);
  bool dummy = false;
  while (!lane_step<true>(tc, t, s, sp, dummy)) {
  }
  lane_exact_result r;
  r.hit = t.hit, r.hit_r = t.hit_r, r.hit_lines = t.hit_lines ? 1 : 0, r.base = s.base;
  return r;
}

// A whole ray in one call (the light-pdf rays of sample_lights_pdf, pt.cpp:1315-1334): closest hit
// against one instance, pushing above whatever the lane's stack already holds (a suspended ray).
YH_DEV hit_t lane_trace(const trace_ctx& tc, lane_stack& s, f3 ro, f3 rd, int first_object) {
  lane_trav t;
  lane_begin(*tc.sc, t, ro, rd, first_object);
  const int sp0  = s.sp;
  bool      redo = false;
  while (!lane_step<false>(tc, t, s, sp0, redo)) {
  }
  if (redo) {
    while (s.sp > sp0) (void)lane_pop(s);
    lane_exact_result r = lane_trace_exact(tc.sc_dev, tc.lds_scene, s.lds, s.ovf, s.sp, s.base, ro, rd, first_object);
    s.base = r.base;
    return lane_hit(tc, r.hit, r.hit_lines != 0, r.hit_r);
  }
  return lane_hit(tc, t.hit, t.hit_lines, t.hit_r);
}

}  // namespace yhd
#endif
