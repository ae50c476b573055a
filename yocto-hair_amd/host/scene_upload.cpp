// scene_upload.cpp — yh_upload_scene = init_bvh + init_lights (pt.cpp:755-818,1695-1740): everything a launch reads is made here, big shapes
// entirely on the device (csrc/bvh_gpu.hip); nothing is deferred to a first launch (rounds 3-5 built the 8- / 16-wide nodes there).
#include "context_internal.h"

#include <atomic>

// The material-only part of eval_hair_brdf (ext.cpp:131-172) plus the
// per-lobe constants the kernels use (dev_hair.h). Same libm as the reference
// (this runs on the host), so these values are bit-identical to what the
// reference recomputes at every hit.
// a plain device allocation owned by `buf`
static int dev_alloc(yh_context* ctx, DevBuf& buf, size_t bytes) {
  buf.reset();
  HIPCHK(ctx, hipMalloc(&buf.p, std::max<size_t>(bytes, 16)));
  buf.bytes = std::max<size_t>(bytes, 16);
  return YH_OK;
}

void make_material(const yh_material& m, yhd_material& d) {
  memset(&d, 0, sizeof(d));
  memcpy(d.emission, m.emission, 12);
  memcpy(d.color, m.color, 12);
  float dmax    = fmax_(fmax_(m.color[0], m.color[1]), m.color[2]);
  d.diffuse_pdf = dmax ? dmax / dmax : 0.0f;  // pt.cpp:456-471 with one lobe
  d.thin        = m.thin;
  d.specular = m.specular, d.metallic = m.metallic, d.roughness = m.roughness, d.ior = m.ior;
  d.transmission = m.transmission;
  d.opacity      = m.opacity * ((1.0f + 1.0f + 1.0f) / 3);  // mean of the {1,1,1} null texture (pt.cpp:425)
  if (d.opacity > 0.999f) d.opacity = 1;
  d.emission_tex = m.emission_tex - 1, d.color_tex = m.color_tex - 1, d.scattering_tex = m.scattering_tex - 1;
  d.trdepth = m.trdepth;
  d.plain = m.specular == 0 && m.metallic == 0 && m.transmission == 0 && d.opacity == 1 && m.emission_tex == 0 &&
            m.color_tex == 0 && m.scattering_tex == 0;
  for (int c = 0; c < 3; c++) {  // reflectivity_to_eta (math.h:4270-4273)
    float r   = fmin_(fmax_(m.color[c], 0.0f), 0.99f);
    d.meta[c] = (1 + std::sqrt(r)) / (1 - std::sqrt(r));
  }
  d.has_volume = !m.thin && m.transmission != 0;
  for (int c = 0; c < 3; c++) {  // eval_vsdf (pt.cpp:520-524)
    d.vol_density[c] = d.has_volume ? -std::log(fmin_(fmax_(m.color[c], 0.0001f), 1.0f)) / m.trdepth : 0.0f;
    d.vol_scatter[c] = m.scattering[c];
  }
  d.vol_anisotropy = m.scanisotropy;
  F3 sa{0, 0, 0};
  if (m.sigma_a[0] || m.sigma_a[1] || m.sigma_a[2]) {
    sa = ld3(m.sigma_a);
  } else if (m.color[0] || m.color[1] || m.color[2]) {  // ext.cpp:121-125
    float bn  = m.beta_n;
    float den = 5.969f - 0.215f * bn + 2.532f * sqr(bn) - 10.73f * powt<3>(bn) + 5.574f * powt<4>(bn) +
                0.245f * powt<5>(bn);
    F3 q = {std::log(m.color[0]) / den, std::log(m.color[1]) / den, std::log(m.color[2]) / den};
    sa   = {q.x * q.x, q.y * q.y, q.z * q.z};
  } else if (m.eumelanin || m.pheomelanin) {  // ext.cpp:115-119
    F3 e = F3{0.419f, 0.697f, 1.37f}, p = F3{0.187f, 0.4f, 1.05f};
    sa   = F3{m.eumelanin * e.x, m.eumelanin * e.y, m.eumelanin * e.z} +
         F3{m.pheomelanin * p.x, m.pheomelanin * p.y, m.pheomelanin * p.z};
  }
  st3(d.sigma_a, sa);
  d.alpha = m.alpha, d.eta = m.eta;
  float bm = m.beta_m, bn = m.beta_n;
  d.v[0] = sqr(0.726f * bm + 0.812f * sqr(bm) + 3.7f * powt<20>(bm));
  d.v[1] = 0.25f * d.v[0];
  d.v[2] = 4 * d.v[0];
  d.v[3] = d.v[2];
  d.s    = 0.626657069f * (0.265f * bn + 1.194f * sqr(bn) + 5.372f * powt<22>(bn));
  d.sin_2k_alpha[0] = std::sin(pif / 180 * d.alpha);
  d.cos_2k_alpha[0] = std::sqrt(fmax_(0.0f, 1 - sqr(d.sin_2k_alpha[0])));
  for (int i = 1; i < 3; i++) {
    d.sin_2k_alpha[i] = 2 * d.cos_2k_alpha[i - 1] * d.sin_2k_alpha[i - 1];
    d.cos_2k_alpha[i] = sqr(d.cos_2k_alpha[i - 1]) - sqr(d.sin_2k_alpha[i - 1]);
  }
  for (int p = 0; p < 4; p++) {
    d.inv_v[p]        = 1 / d.v[p];
    d.log_inv_2v[p]   = std::log(1 / (2 * d.v[p]));
    d.exp_m2_inv_v[p] = std::exp(-2 / d.v[p]);
    d.mp_den[p]       = ::sinh((double)(1 / d.v[p])) * 2 * d.v[p];
  }
  float cb   = 1 / (1 + std::exp(-pif / d.s));
  float ca   = 1 / (1 + std::exp(-(-pif) / d.s));
  d.tl_cdf_a = ca;
  d.tl_norm  = cb - ca;
}


int yh_upload_scene(yh_context* ctx, const yh_scene_desc* sd) {
  if (!ctx) return YH_E_INVALID;
  if (!sd) return fail(ctx, YH_E_INVALID, "scene is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->poisoned) return fail(ctx, YH_E_DEVICE, "a launch of this context exceeded its deadline: the context refuses further work, destroy it");
  YH_WAIT(ctx);  // (an asynchronous launch may still be reading the scene this call replaces: wait for it, within the deadline)
  if (sd->num_objects <= 0) return fail(ctx, YH_E_INVALID, "scene has no objects");
  if (sd->num_environments > YH_MAX_ENVS) return fail(ctx, YH_E_INVALID, "more than %d environments", YH_MAX_ENVS);
  // YHAIR_TIMING=1: stage times of the upload on stderr
  const bool timing = getenv("YHAIR_TIMING") && atoi(getenv("YHAIR_TIMING")) != 0;
  auto       t_last = std::chrono::steady_clock::now();
  auto       lap    = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[yhair] upload: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // ---- per-shape BVHs and leaf-ordered records ---------------------------------------------------------------------------
  // BIG shapes (>= 32 768 primitives: the hair) never leave the device (round 6): their vertex arrays cross PCIe once, as they are, and
  // bounds, the reference's tree (csrc/bvh_gpu.hip), the leaf-ordered records and — below — the wide collapses are made there. SMALL shapes
  // (the configs' sphere and lights: 10 ms together) are built on the host as before and their trees and records uploaded into the same arrays;
  // the collapses are the device's for both. YHAIR_BVH=host: every shape the small way.
  struct ShapeInfo {
    int kind, prim_base, vert_base, elem_base, has_normals, depth;
    int depth8, depth16;  // depths of the same tree collapsed three / four levels at a time
    yhh::Box root;
    int num_prims;
    int num_nodes = 0, levels = 1;   // the binary tree on the device: node count, levels, first node of every level
    int level_first[130] = {0};
    int wide_count[3] = {0, 0, 0};   // its 4- / 8- / 16-wide nodes
    std::vector<yhd_float4> host_prims;  // SMALL shapes: their leaf records on the host too (the LDS light table is made from them)
  };
  std::vector<ShapeInfo>  info(sd->num_shapes);
  std::vector<DevBuf>     d_tree(sd->num_shapes);  // binary nodes per shape (8 floats each), until the collapses are made
  std::vector<yhd_float4> vpos;
  std::vector<float>      vtex;  // 2 per vertex, zeros for shapes without texture coordinates
  std::vector<yhd_int4>   elems;
  size_t total_prim_f4 = 0;
  for (int si = 0; si < sd->num_shapes; si++) {
    auto& s = sd->shapes[si];
    if (s.num_vertices <= 0 || !s.positions) return fail(ctx, YH_E_INVALID, "shape %d has no vertices", si);
    const bool lines = s.num_lines > 0;
    if (!lines && s.num_triangles <= 0) return fail(ctx, YH_E_INVALID, "shape %d has no lines or triangles", si);
    const int nel = lines ? s.num_lines : s.num_triangles;
    // a leaf reference packs its first record into 27 bits (host/bvh_build.cpp: count << 27 | start)
    if (nel >= (1 << 27)) return fail(ctx, YH_E_INVALID, "shape %d has %d elements (limit %d)", si, nel, (1 << 27) - 1);
    info[si].kind = lines ? YH_KIND_LINES : YH_KIND_TRIANGLES, info[si].num_prims = nel, info[si].has_normals = s.normals != nullptr;
    info[si].prim_base = (int)total_prim_f4;
    total_prim_f4 += (size_t)nel * (lines ? 4 : 6);
  }
  if (total_prim_f4 > (size_t)std::numeric_limits<int>::max()) return fail(ctx, YH_E_INVALID, "scene too large for 32-bit record offsets (%zu primitive float4)", total_prim_f4);
  int rc;
  // From here on the previous scene's device arrays are being replaced: a call that fails half-way must leave a context WITHOUT a scene (every launch
  // then says "before yh_upload_scene"), never one whose scene table points at freed arrays.
  ctx->have_scene = false, ctx->have_state = false;
  ctx->scene.lane_blob = nullptr, ctx->scene.prims = nullptr;
  ctx->d_prims.reset(), ctx->d_lane_blob.reset();  // (nothing of this context is running: waited for above)
  if ((rc = dev_alloc(ctx, ctx->d_prims, total_prim_f4 * 16))) return rc;
  lap("validation, record array");
  const char* bvh_env   = getenv("YHAIR_BVH");  // developer switch, read per upload: "host" = every shape the small way, "device" = every shape on the device
  const bool  host_only = bvh_env && !strcmp(bvh_env, "host"), device_all = bvh_env && !strcmp(bvh_env, "device");
  for (int si = 0; si < sd->num_shapes; si++) {
    auto&      s     = sd->shapes[si];
    auto&      I     = info[si];
    const bool lines = I.kind == YH_KIND_LINES;
    const int  nel   = I.num_prims;
    const int* idx   = lines ? s.lines : s.triangles;
    {  // (a parallel pass: 3.2 M indices of a hair model)
      std::atomic<bool> bad{false};
      parallel_for(nel * (lines ? 2 : 3), [&](int k) {
        if (idx[k] < 0 || idx[k] >= s.num_vertices) bad = true;
      });
      if (bad) return fail(ctx, YH_E_INVALID, "shape %d: vertex index out of range", si);
    }
    I.vert_base = (int)vpos.size(), I.elem_base = (int)elems.size();
    yhd_float4* d_recs = (yhd_float4*)ctx->d_prims.p + I.prim_base;
    if ((nel >= 32768 || device_all) && !host_only) {
      // ---- on the device ----
      const size_t nv = (size_t)s.num_vertices;
      DevBuf d_pos, d_nrm, d_rad, d_idx, d_boxes, d_pid;
      auto h2d = [&](DevBuf& buf, const void* src, size_t bytes) -> int {
        if (int arc = dev_alloc(ctx, buf, bytes)) return arc;
        HIPCHK(ctx, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return YH_OK;
      };
      if ((rc = h2d(d_pos, s.positions, nv * 12))) return rc;
      if (s.radius && (rc = h2d(d_rad, s.radius, nv * 4))) return rc;
      if ((rc = h2d(d_idx, idx, (size_t)nel * (lines ? 8 : 12)))) return rc;
      if ((rc = dev_alloc(ctx, d_boxes, (size_t)nel * 24))) return rc;
      if ((rc = dev_alloc(ctx, d_pid, (size_t)nel * 4))) return rc;
      if ((rc = dev_alloc(ctx, d_tree[si], ((size_t)2 * nel + 1) * 32))) return rc;
      int e = yhk_prim_boxes(lines ? 1 : 0, nel, (const float*)d_pos.p, (const float*)d_rad.p, (const int*)d_idx.p, (float*)d_boxes.p, ctx->stream);
      if (e) return fail(ctx, YH_E_DEVICE, "primitive bounds: %s", hipGetErrorString((hipError_t)e));
      // (the normals' copy is queued behind the kernels that do not need it: it travels while the tree is built)
      if (s.normals && (rc = h2d(d_nrm, s.normals, nv * 12))) return rc;
      e = yhk_bvh_build_resident(nel, (const float*)d_boxes.p, (float*)d_tree[si].p, (int*)d_pid.p, &I.num_nodes, &I.levels, I.level_first, ctx->stream);
      if (e) return fail(ctx, YH_E_DEVICE, "device BVH build: %s", hipGetErrorString((hipError_t)e));
      lap("device: bounds + reference tree");
      e = yhk_leaf_records(lines ? 1 : 0, nel, (const int*)d_pid.p, (const float*)d_pos.p, (const float*)d_nrm.p, (const float*)d_rad.p, (const int*)d_idx.p, d_recs, ctx->stream);
      if (e) return fail(ctx, YH_E_DEVICE, "leaf records: %s", hipGetErrorString((hipError_t)e));
      float root8[8];
      HIPCHK(ctx, hipMemcpyAsync(root8, d_tree[si].p, 32, hipMemcpyDeviceToHost, ctx->stream));
      YH_WAIT(ctx);  // (the vertex arrays go out of scope)
      memcpy(I.root.min, root8, 12), memcpy(I.root.max, root8 + 3, 12);
      lap("device: leaf records");
    } else {
      // ---- on the host (small shapes) ----
      auto pos = [&](int v) { return ld3(s.positions + 3 * (size_t)v); };
      auto rad = [&](int v) { return s.radius ? s.radius[v] : 0.001f; };  // add_radius, sceneio.cpp:390
      std::vector<yhh::Box> boxes(nel);
      parallel_for(nel, [&](int e) {
        if (lines) {  // line_bounds (math.h:3037-3040)
          int a = idx[2 * e], b = idx[2 * e + 1];
          F3  p0 = pos(a), p1 = pos(b);
          float r0 = rad(a), r1 = rad(b);
          float lo0[3] = {p0.x - r0, p0.y - r0, p0.z - r0}, lo1[3] = {p1.x - r1, p1.y - r1, p1.z - r1};
          float hi0[3] = {p0.x + r0, p0.y + r0, p0.z + r0}, hi1[3] = {p1.x + r1, p1.y + r1, p1.z + r1};
          for (int k = 0; k < 3; k++) boxes[e].min[k] = fmin_(lo0[k], lo1[k]), boxes[e].max[k] = fmax_(hi0[k], hi1[k]);
        } else {  // triangle_bounds (math.h:3041-3044)
          const float* p0 = s.positions + 3 * (size_t)idx[3 * e];
          const float* p1 = s.positions + 3 * (size_t)idx[3 * e + 1];
          const float* p2 = s.positions + 3 * (size_t)idx[3 * e + 2];
          for (int k = 0; k < 3; k++) {
            boxes[e].min[k] = fmin_(p0[k], fmin_(p1[k], p2[k]));
            boxes[e].max[k] = fmax_(p0[k], fmax_(p1[k], p2[k]));
          }
        }
      });
      yhh::Tree tree;
      yhh::build_bvh(tree, boxes);
      // the tree as the device builder leaves it: 8 floats per node (yhh::Node has that layout byte for byte), the first node of every level
      static_assert(sizeof(yhh::Node) == 32 && offsetof(yhh::Node, start) == 24 && offsetof(yhh::Node, num) == 28 && offsetof(yhh::Node, internal) == 30 && offsetof(yhh::Node, axis) == 31,
          "yhh::Node is the device's node record");
      {
        std::vector<int> level(tree.nodes.size(), 0);
        for (size_t n = 0; n < tree.nodes.size(); n++)
          if (tree.nodes[n].internal) level[(size_t)tree.nodes[n].start] = level[(size_t)tree.nodes[n].start + 1] = level[n] + 1;
        I.num_nodes = (int)tree.nodes.size(), I.levels = level.empty() ? 1 : level.back() + 1;  // (breadth-first numbering: levels are contiguous, the last node is on the last one)
        if (I.levels > 128) return fail(ctx, YH_E_INVALID, "shape %d: tree of %d levels", si, I.levels);
        for (int l = 0; l <= I.levels; l++) I.level_first[l] = I.num_nodes;
        for (size_t n = tree.nodes.size(); n-- > 0;) I.level_first[level[n]] = (int)n;
      }
      if ((rc = upload(ctx, d_tree[si], tree.nodes.data(), tree.nodes.size() * 32))) return rc;
      I.root = tree.nodes[0].bbox;
      auto nrm = [&](int v) { return s.normals ? ld3(s.normals + 3 * (size_t)v) : F3{0, 0, 0}; };
      const size_t per = lines ? 4 : 6;
      I.host_prims.resize(per * (size_t)nel);
      yhd_float4* out = I.host_prims.data();
      parallel_for(nel, [&](int slot) {
        int   e = tree.primitives[slot];
        float ew;
        memcpy(&ew, &e, 4);
        yhd_float4* r = out + per * (size_t)slot;
        if (lines) {
          int a = idx[2 * e], b = idx[2 * e + 1];
          F3  p0 = pos(a), p1 = pos(b), t0 = nrm(a), t1 = nrm(b);
          r[0] = {p0.x, p0.y, p0.z, rad(a)}, r[1] = {p1.x, p1.y, p1.z, rad(b)};
          r[2] = {t0.x, t0.y, t0.z, ew}, r[3] = {t1.x, t1.y, t1.z, 0};
        } else {
          int a = idx[3 * e], b = idx[3 * e + 1], cc = idx[3 * e + 2];
          F3  p0 = pos(a), p1 = pos(b), p2 = pos(cc), n0 = nrm(a), n1 = nrm(b), n2 = nrm(cc);
          r[0] = {p0.x, p0.y, p0.z, ew}, r[1] = {p1.x, p1.y, p1.z, 0}, r[2] = {p2.x, p2.y, p2.z, 0};
          r[3] = {n0.x, n0.y, n0.z, 0}, r[4] = {n1.x, n1.y, n1.z, 0}, r[5] = {n2.x, n2.y, n2.z, 0};
        }
      });
      HIPCHK(ctx, hipMemcpy(d_recs, I.host_prims.data(), I.host_prims.size() * 16, hipMemcpyHostToDevice));
      lap("host: small shape");
    }
    {  // depths of the wide trees: a W-wide node stands for every internal binary node at a level that is a multiple of log2 W, so the wide
       // depth is 1 + deepest internal level / log2 W (what collapse_wide / _wide8 / _wide16 return). The last level holds leaves only.
      const int deepest = std::max(0, I.levels - 2);
      I.depth = 1 + deepest / 2, I.depth8 = 1 + deepest / 3, I.depth16 = 1 + deepest / 4;
    }
    {
      // Per-vertex positions and per-element indices are read on the device only to sample a point
      // on an area light (triangles, pt.cpp:1287-1292) and to interpolate texture coordinates; the
      // traversal and the shading of a hit use the leaf records. Hair without texture coordinates —
      // nearly all of a scene's bytes — therefore has no entry in these arrays.
      const bool per_vertex = !lines || s.texcoords != nullptr;
      if (!per_vertex) {
        I.vert_base = 0, I.elem_base = 0;
      } else {
        auto pos = [&](int v) { return ld3(s.positions + 3 * (size_t)v); };
        const size_t at = vpos.size();
        vpos.resize(at + (size_t)s.num_vertices);
        vtex.resize(2 * (at + (size_t)s.num_vertices), 0.0f);
        if (s.texcoords) memcpy(&vtex[2 * at], s.texcoords, sizeof(float) * 2 * (size_t)s.num_vertices);
        parallel_for(s.num_vertices, [&](int v) {
          F3 p = pos(v);
          vpos[at + (size_t)v] = {p.x, p.y, p.z, lines ? (s.radius ? s.radius[v] : 0.001f) : 0.0f};
        });
        const size_t ea = elems.size();
        elems.resize(ea + (size_t)nel);
        parallel_for(nel, [&](int e) {
          elems[ea + (size_t)e] = lines ? yhd_int4{idx[2 * e], idx[2 * e + 1], 0, 0}
                                        : yhd_int4{idx[3 * e], idx[3 * e + 1], idx[3 * e + 2], 0};
        });
      }
    }
  }
  // ---- the wide collapses: the index of every wide node (one flag pass + one scan per width), then the layout of the ONE array the traversal
  // kernels read (yh_device.h: lane_blob, 32-byte units): [test records of every shape][4-wide nodes][8-wide nodes][16-wide nodes] ----
  std::vector<DevBuf> d_wflag((size_t)sd->num_shapes * 3), d_widx((size_t)sd->num_shapes * 3);
  for (int si = 0; si < sd->num_shapes; si++)
    for (int w = 0; w < 3; w++) {
      auto&  I = info[si];
      DevBuf &f = d_wflag[(size_t)si * 3 + w], &x = d_widx[(size_t)si * 3 + w];
      if ((rc = dev_alloc(ctx, f, ((size_t)I.num_nodes + 1) * 4))) return rc;
      if ((rc = dev_alloc(ctx, x, ((size_t)I.num_nodes + 1) * 4))) return rc;
      int e = yhk_wide_index(I.num_nodes, (const float*)d_tree[si].p, I.levels, I.level_first, 2 + w, (unsigned int*)f.p, (unsigned int*)x.p, &I.wide_count[w], ctx->stream);
      if (e) return fail(ctx, YH_E_DEVICE, "wide-node index: %s", hipGetErrorString((hipError_t)e));
    }
  ctx->lane_shapes.assign((size_t)sd->num_shapes, yh_context::LaneShape{});
  long long U8 = 0, U16 = 0, blob_units = 0;
  {
    long long at = 0;
    for (int si = 0; si < sd->num_shapes; si++) {
      auto& L = ctx->lane_shapes[(size_t)si];
      L.kind = info[si].kind, L.num_nodes = info[si].wide_count[0], L.prim_base = info[si].prim_base, L.num_prims = info[si].num_prims;
      L.test_off = at, at += (long long)L.num_prims * (L.kind == YH_KIND_LINES ? 1 : 2);
    }
    at = (at + 3) / 4 * 4 + 4;  // (nodes on 128-byte lines; four units of slack behind the last test record: a leaf step reads 64 bytes)
    if (at >= (1ll << 27)) return fail(ctx, YH_E_INVALID, "scene too large for 27-bit leaf offsets (%lld test-record units)", at);
    for (int si = 0; si < sd->num_shapes; si++) ctx->lane_shapes[(size_t)si].node_off = at, at += 4ll * info[si].wide_count[0];
    ctx->lane_units = at + 4;  // what the one-lane kernels address (32-bit byte offsets: launch_plan.cpp: lane_kernels_can_address)
    U8 = (ctx->lane_units + 3) / 4 * 4, at = U8;
    for (int si = 0; si < sd->num_shapes; si++) ctx->lane_shapes[(size_t)si].node_off8 = at, at += 8ll * info[si].wide_count[1];
    U16 = at;
    for (int si = 0; si < sd->num_shapes; si++) ctx->lane_shapes[(size_t)si].node_off16 = at, at += 16ll * info[si].wide_count[2];
    blob_units = at + 4;
    if (blob_units >= (1ll << 30)) return fail(ctx, YH_E_INVALID, "scene too large for 30-bit node offsets (%lld units)", blob_units);
    (void)U16;
  }
  if ((rc = alloc_zero(ctx, ctx->d_lane_blob, (size_t)blob_units * 32))) return rc;
  for (int si = 0; si < sd->num_shapes; si++) {
    auto& L = ctx->lane_shapes[(size_t)si];
    auto& I = info[si];
    int e = yhk_lane_tests((const yhd_float4*)ctx->d_prims.p, (yhd_float4*)ctx->d_lane_blob.p, L.kind, L.prim_base, L.num_prims, L.test_off, ctx->stream);
    const long long offs[3] = {L.node_off, L.node_off8, L.node_off16};
    for (int w = 0; w < 3 && !e; w++)
      e = yhk_wide_collapse(2 + w, I.num_nodes, (const float*)d_tree[si].p, (const unsigned int*)d_wflag[(size_t)si * 3 + w].p, (const unsigned int*)d_widx[(size_t)si * 3 + w].p,
          L.kind == YH_KIND_LINES ? 1 : 0, offs[w], L.test_off, ctx->d_lane_blob.p, ctx->stream);
    if (e) return fail(ctx, YH_E_DEVICE, "wide collapse: %s", hipGetErrorString((hipError_t)e));
  }
  YH_WAIT(ctx);
  d_tree.clear(), d_wflag.clear(), d_widx.clear();
  lap("device: wide collapses, blob");
  // ---- objects and the scene-level BVH (pt.cpp:792-814) -------------------
  std::vector<yhd_object> objects(sd->num_objects);
  std::vector<yhh::Box>   obj_boxes(sd->num_objects);
  for (int oi = 0; oi < sd->num_objects; oi++) {
    auto& o = sd->objects[oi];
    if (o.shape < 0 || o.shape >= sd->num_shapes || o.material < 0 || o.material >= sd->num_materials)
      return fail(ctx, YH_E_INVALID, "object %d references a missing shape or material", oi);
    auto& I = info[o.shape];
    auto& d = objects[oi];
    memcpy(d.frame, o.frame, 48);
    inverse_frame(o.frame, true, d.inv_frame);
    d.kind = I.kind, d.node_base = 0, d.prim_base = I.prim_base, d.vert_base = I.vert_base;
    d.elem_base = I.elem_base, d.has_normals = I.has_normals, d.material = o.material, d.has_texcoords = sd->shapes[o.shape].texcoords != nullptr;
    const auto& LS = ctx->lane_shapes[(size_t)o.shape];
    d.lane_root = (int)LS.node_off, d.lane_test = (int)LS.test_off, d.lane_root8 = (int)LS.node_off8, d.lane_root16 = (int)LS.node_off16;
    // transform_bbox (math.h:3174-3185)
    const yhh::Box& b = I.root;
    float lo[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float hi[3] = {std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(),
        std::numeric_limits<float>::lowest()};
    for (int c = 0; c < 8; c++) {
      F3 corner = {(c & 4) ? b.max[0] : b.min[0], (c & 2) ? b.max[1] : b.min[1], (c & 1) ? b.max[2] : b.min[2]};
      F3 t      = transform_point(o.frame, corner);
      float tv[3] = {t.x, t.y, t.z};
      for (int k = 0; k < 3; k++) lo[k] = fmin_(lo[k], tv[k]), hi[k] = fmax_(hi[k], tv[k]);
    }
    for (int k = 0; k < 3; k++) obj_boxes[oi].min[k] = lo[k], obj_boxes[oi].max[k] = hi[k];
    {  // the same box with a margin a thousand times the rounding of either box test
      float ext = fmax_(fmax_(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
      float eps = 1e-3f * ext + 1e-5f;
      for (int k = 0; k < 3; k++) d.wbox_min[k] = lo[k] - eps, d.wbox_max[k] = hi[k] + eps;
      d.wbox_min[3] = d.wbox_max[3] = 0;
    }
  }
  // array offsets on the device are 32-bit float4 indices
  if (vpos.size() > (size_t)std::numeric_limits<int>::max()) return fail(ctx, YH_E_INVALID, "scene too large for 32-bit vertex offsets (%zu)", vpos.size());
  yhh::Tree scene_tree;
  yhh::build_bvh(scene_tree, obj_boxes);
  std::vector<yhd_float4> scene_nodes;
  for (auto& n : scene_tree.nodes) scene_nodes.push_back(node_lo(n)), scene_nodes.push_back(node_hi(n));
  int max_shape_depth = 0;
  for (auto& I : info) max_shape_depth = std::max(max_shape_depth, I.depth);
  // a wide node pushes at most three entries and keeps the fourth in a register
  ctx->stack_need = scene_tree.max_depth + 4 + 3 * max_shape_depth + 2;
  int max_shape_depth8 = 0;
  for (auto& I : info) max_shape_depth8 = std::max(max_shape_depth8, I.depth8);
  ctx->stack_need8 = scene_tree.max_depth + 4 + 7 * max_shape_depth8 + 2;  // an 8-wide node pushes at most seven
  int max_shape_depth16 = 0;
  for (auto& I : info) max_shape_depth16 = std::max(max_shape_depth16, I.depth16);
  ctx->stack_need16 = scene_tree.max_depth + 4 + 15 * max_shape_depth16 + 2;
  if (ctx->stack_need > yhk_stack_entries())
    return fail(ctx, YH_E_INVALID, "BVH too deep for the traversal stack (%d > %d)", ctx->stack_need, yhk_stack_entries());
  // ---- materials ---------------------------------------------------------
  std::vector<yhd_material> materials(sd->num_materials);
  int general_materials = 0;
  for (int i = 0; i < sd->num_materials; i++) {
    make_material(sd->materials[i], materials[i]);
    if (!materials[i].plain) general_materials = 1;
  }
  // ---- lights (pt.cpp:1695-1740) -----------------------------------------
  yhd_scene sc{};
  std::vector<float>      light_cdf;
  std::vector<yhd_float4> env_texels;
  std::vector<int>        small_lights;  // lights whose record goes into the LDS light table
  for (int oi = 0; oi < sd->num_objects; oi++) {
    auto& o = sd->objects[oi];
    auto& m = sd->materials[o.material];
    if (m.emission[0] == 0 && m.emission[1] == 0 && m.emission[2] == 0) continue;
    auto& s = sd->shapes[o.shape];
    if (s.num_lines > 0 || s.num_triangles <= 0) continue;
    if (sc.num_lights >= YH_MAX_LIGHTS) return fail(ctx, YH_E_INVALID, "more than %d lights", YH_MAX_LIGHTS);
    auto& L = sc.lights[sc.num_lights++];
    L.object = oi, L.environment = -1, L.cdf_base = (int)light_cdf.size(), L.cdf_count = s.num_triangles, L.small_base = -1;
    if (s.num_triangles <= YH_SMALL_LIGHT_TRIS) small_lights.push_back(sc.num_lights - 1);  // its record is made below, once the cdf exists
    else general_materials = 1;  // a light sampled and intersected through memory: the general kernel variant (dev_path.h: BIG_LIGHTS)
    for (int t = 0; t < s.num_triangles; t++) {
      F3 p0 = ld3(s.positions + 3 * (size_t)s.triangles[3 * t]), p1 = ld3(s.positions + 3 * (size_t)s.triangles[3 * t + 1]),
         p2 = ld3(s.positions + 3 * (size_t)s.triangles[3 * t + 2]);
      F3    c    = cross(p1 - p0, p2 - p0);
      float area = std::sqrt(dot(c, c)) / 2;  // triangle_area (math.h:3306)
      if (t) area += light_cdf.back();
      light_cdf.push_back(area);
    }
  }
  sc.num_environments = sd->num_environments;
  for (int ei = 0; ei < sd->num_environments; ei++) {
    auto& e = sd->environments[ei];
    auto& d = sc.environments[ei];
    memcpy(d.frame, e.frame, 48);
    inverse_frame(e.frame, false, d.inv_frame);
    memcpy(d.emission, e.emission, 12);
    d.tex_w = e.texels ? e.tex_width : 0, d.tex_h = e.texels ? e.tex_height : 0;
    d.texel_base = (int)env_texels.size();
    if (e.texels)
      for (size_t t = 0; t < (size_t)e.tex_width * e.tex_height; t++)
        env_texels.push_back({e.texels[3 * t], e.texels[3 * t + 1], e.texels[3 * t + 2], 0});
    if (e.emission[0] == 0 && e.emission[1] == 0 && e.emission[2] == 0) continue;
    if (sc.num_lights >= YH_MAX_LIGHTS) return fail(ctx, YH_E_INVALID, "more than %d lights", YH_MAX_LIGHTS);
    auto& L = sc.lights[sc.num_lights++];
    L.object = -1, L.environment = ei, L.cdf_base = (int)light_cdf.size(), L.cdf_count = 0, L.small_base = -1;
    if (e.texels) {
      size_t n    = (size_t)e.tex_width * e.tex_height;
      L.cdf_count = (int)n;
      for (size_t i = 0; i < n; i++) {
        int   iy    = (int)(i / e.tex_width);
        float th    = (iy + 0.5f) * pif / e.tex_height;
        float mx    = fmax_(fmax_(e.texels[3 * i], e.texels[3 * i + 1]), e.texels[3 * i + 2]);
        float value = mx * std::sin(th);
        if (i) value += light_cdf.back();
        light_cdf.push_back(value);
      }
    }
  }
  if (sc.num_lights == 0) return fail(ctx, YH_E_INVALID, "scene has no lights (the path sampler needs at least one)");
  // ---- tables the kernels keep in LDS (yh_device.h) -----------------------------------------------------
  // small area lights: root box, leaf-ordered triangles, area cdf — everything sample_lights / sample_lights_pdf read
  std::vector<yhd_float4> light_table;
  for (int li : small_lights) {
    auto& L  = sc.lights[li];
    auto& I  = info[sd->objects[L.object].shape];
    if (I.host_prims.empty()) {  // (a shape made on the device — YHAIR_BVH=device: a light of <= 4 triangles is otherwise a small shape): its few records back
      I.host_prims.resize((size_t)6 * L.cdf_count);
      HIPCHK(ctx, hipMemcpy(I.host_prims.data(), (const yhd_float4*)ctx->d_prims.p + I.prim_base, I.host_prims.size() * 16, hipMemcpyDeviceToHost));
    }
    const yhd_float4* rec = I.host_prims.data();
    L.small_base = (int)light_table.size();
    yhd_float4 b0{I.root.min[0], I.root.min[1], I.root.min[2], 0}, b1{I.root.max[0], I.root.max[1], I.root.max[2], light_cdf[(size_t)L.cdf_base + L.cdf_count - 1]};
    memcpy(&b0.w, &L.cdf_count, 4);
    light_table.push_back(b0), light_table.push_back(b1);
    for (int t = 0; t < YH_SMALL_LIGHT_TRIS; t++)
      for (int k = 0; k < 3; k++) light_table.push_back(t < L.cdf_count ? rec[6 * t + k] : yhd_float4{0, 0, 0, 0});
    yhd_float4 cdf{0, 0, 0, 0};
    for (int t = 0; t < L.cdf_count; t++) (&cdf.x)[t] = light_cdf[(size_t)L.cdf_base + t];
    light_table.push_back(cdf);
  }
  // coarse index of the first textured environment light's cdf: 2048 entries (8 KB) halve the dependent fetches of
  // its 21-step binary search
  std::vector<float> env_tab;
  sc.env_tab_light = -1, sc.env_tab_k = 0, sc.env_tab_stride = 0;
  for (int li = 0; li < sc.num_lights && sc.env_tab_light < 0; li++) {
    auto& L = sc.lights[li];
    if (L.environment < 0 || L.cdf_count < 4096) continue;
    const int n = L.cdf_count, S = (n + 2047) / 2048, K = (n + S - 1) / S;
    env_tab.resize((size_t)K);
    for (int k = 0; k < K; k++) env_tab[(size_t)k] = light_cdf[(size_t)L.cdf_base + (size_t)std::min<int64_t>(n, (int64_t)(k + 1) * S) - 1];
    sc.env_tab_light = li, sc.env_tab_k = K, sc.env_tab_stride = S;
  }
  // ---- material colour textures (lookup_texture's per-texel conversion done once, pt.cpp:147-164) --------
  std::vector<yhd_texture> textures((size_t)std::max(0, sd->num_textures));
  std::vector<yhd_float4>  tex_texels;
  {
    std::vector<char> need_linear(textures.size(), 0);
    for (int i = 0; i < sd->num_materials; i++) {
      auto& m = sd->materials[i];
      for (int id : {m.emission_tex, m.color_tex, m.scattering_tex})
        if (id < 0 || id > sd->num_textures) return fail(ctx, YH_E_INVALID, "material %d references a missing texture", i);
      if (m.emission_tex > 0) need_linear[(size_t)m.emission_tex - 1] = 1;  // transmission *= emission_tex.x, linear (pt.cpp:421)
    }
    auto srgb_to_rgb = [](float srgb) {  // math.h:3742-3745
      return (srgb <= 0.04045) ? srgb / 12.92f : std::pow((srgb + 0.055f) / (1.0f + 0.055f), 2.4f);
    };
    for (size_t t = 0; t < textures.size(); t++) {
      auto& src = sd->textures[t];
      if (src.width <= 0 || src.height <= 0 || !src.pixels) return fail(ctx, YH_E_INVALID, "texture %d is empty", (int)t);
      size_t n = (size_t)src.width * src.height;
      auto&  d = textures[t];
      d.width = src.width, d.height = src.height, d.srgb_base = (int)tex_texels.size(), d.linear_base = -1;
      tex_texels.resize(tex_texels.size() + n);
      yhd_float4* out = tex_texels.data() + d.srgb_base;
      if (src.is_byte) {
        auto b = (const unsigned char*)src.pixels;
        parallel_for((int)n, [&](int i) {
          out[i] = {srgb_to_rgb(b[3 * (size_t)i] / 255.0f), srgb_to_rgb(b[3 * (size_t)i + 1] / 255.0f),
              srgb_to_rgb(b[3 * (size_t)i + 2] / 255.0f), 0};
        });
        if (need_linear[t]) {
          d.linear_base = (int)tex_texels.size();
          tex_texels.resize(tex_texels.size() + n);
          yhd_float4* lin = tex_texels.data() + d.linear_base;
          parallel_for((int)n, [&](int i) { lin[i] = {b[3 * (size_t)i] / 255.0f, b[3 * (size_t)i + 1] / 255.0f, b[3 * (size_t)i + 2] / 255.0f, 0}; });
        }
      } else {
        auto f = (const float*)src.pixels;
        parallel_for((int)n, [&](int i) { out[i] = {f[3 * (size_t)i], f[3 * (size_t)i + 1], f[3 * (size_t)i + 2], 0}; });
        d.linear_base = d.srgb_base;
      }
    }
  }
  lap("objects, materials, lights");
  // ---- upload ------------------------------------------------------------
  if ((rc = upload(ctx, ctx->d_vpos, vpos.data(), vpos.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_elems, elems.data(), elems.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_objects, objects.data(), objects.size() * sizeof(yhd_object)))) return rc;
  if ((rc = upload(ctx, ctx->d_materials, materials.data(), materials.size() * sizeof(yhd_material)))) return rc;
  if ((rc = upload(ctx, ctx->d_scene_nodes, scene_nodes.data(), scene_nodes.size() * 16))) return rc;
  std::vector<int> scene_prims_padded = scene_tree.primitives;
  scene_prims_padded.resize((scene_prims_padded.size() + 3) / 4 * 4, 0);  // staged to LDS as float4
  if ((rc = upload(ctx, ctx->d_scene_prims, scene_prims_padded.data(), scene_prims_padded.size() * 4))) return rc;
  if ((rc = upload(ctx, ctx->d_light_cdf, light_cdf.data(), light_cdf.size() * 4))) return rc;
  if ((rc = upload(ctx, ctx->d_light_table, light_table.data(), light_table.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_env_tab, env_tab.data(), env_tab.size() * 4))) return rc;
  if ((rc = upload(ctx, ctx->d_env_texels, env_texels.data(), env_texels.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_textures, textures.data(), textures.size() * sizeof(yhd_texture)))) return rc;
  if ((rc = upload(ctx, ctx->d_tex_texels, tex_texels.data(), tex_texels.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_vtex, vtex.data(), vtex.size() * 4))) return rc;
  lap("hipMalloc + H2D copies");
  sc.nodes = nullptr, sc.prims = (const yhd_float4*)ctx->d_prims.p;  // (the 4-wide node array of rounds 1-5 exists only inside the lane blob)
  sc.vpos = (const yhd_float4*)ctx->d_vpos.p;
  sc.elems = (const yhd_int4*)ctx->d_elems.p;
  sc.objects = (const yhd_object*)ctx->d_objects.p, sc.materials = (const yhd_material*)ctx->d_materials.p;
  sc.scene_nodes = (const yhd_float4*)ctx->d_scene_nodes.p, sc.scene_prims = (const int*)ctx->d_scene_prims.p;
  sc.num_scene_nodes = (int)scene_tree.nodes.size(), sc.num_objects = sd->num_objects;
  sc.light_cdf = (const float*)ctx->d_light_cdf.p, sc.env_texels = (const yhd_float4*)ctx->d_env_texels.p;
  sc.light_table = (const yhd_float4*)ctx->d_light_table.p, sc.light_table_f4 = (int)light_table.size();
  sc.env_tab = (const float*)ctx->d_env_tab.p;
  sc.stack_entries = std::max(8, (ctx->stack_need + 7) / 8 * 8);
  sc.stack_entries8 = std::max(8, (ctx->stack_need8 + 7) / 8 * 8);
  sc.stack_entries16 = std::max(8, (ctx->stack_need16 + 7) / 8 * 8);
  sc.lane_blob = (const yhd_float4*)ctx->d_lane_blob.p, sc.lane_blob_units = blob_units;
  sc.textures = (const yhd_texture*)ctx->d_textures.p, sc.tex_texels = (const yhd_float4*)ctx->d_tex_texels.p;
  sc.vtex = (const float*)ctx->d_vtex.p;
  memcpy(sc.camera.frame, sd->camera.frame, 48);
  sc.camera.lens = sd->camera.lens, sc.camera.film_x = sd->camera.film[0], sc.camera.film_y = sd->camera.film[1];
  sc.camera.focus = sd->camera.focus, sc.camera.aperture = sd->camera.aperture;
  sc.num_nodes_total = 0, sc.num_prim_f4 = (int)total_prim_f4;
  for (auto& I : info) sc.num_nodes_total += I.wide_count[0];
  sc.general_materials = general_materials;
  {  // scene-level LDS table: objects (YH_OBJECT_F4 = 11 float4 each), scene BVH nodes (2 float4 each), primitive ids; up to 10 KB = 46
     // objects (a scene with more runs the GENERAL kernel variants, which read the table from memory)
    static_assert(sizeof(yhd_object) == 16 * YH_OBJECT_F4, "yhd_object is staged to LDS as float4");
    int f4 = YH_OBJECT_F4 * sd->num_objects + 2 * (int)scene_tree.nodes.size() + (sd->num_objects + 3) / 4;
    sc.lds_scene_f4 = f4 * 16 <= 10240 ? f4 : 0;
  }
  {  // the material table in LDS; the plain kernel variants rely on it and on the scene-level table (dev_path.h)
    static_assert(sizeof(yhd_material) == 16 * YH_MATERIAL_F4, "yhd_material is staged to LDS as float4");
    sc.lds_materials = sd->num_materials <= 24 ? sd->num_materials : 0;
    if (sc.lds_materials == 0 || sc.lds_scene_f4 == 0) sc.general_materials = 1;
  }
  ctx->scene      = sc;
  {  // fingerprint of the scene for the process-wide trial record: counts, camera, materials, objects, a sample of the geometry
    uint64_t h = 1469598103934665603ULL;
    auto mix = [&](const void* p, size_t n) {
      const unsigned char* b = (const unsigned char*)p;
      for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ULL;
    };
    mix(&sd->camera, sizeof(sd->camera));
    mix(sd->materials, sizeof(yh_material) * (size_t)sd->num_materials);
    mix(sd->objects, sizeof(yh_object) * (size_t)sd->num_objects);
    for (int i = 0; i < sd->num_shapes; i++) {
      const yh_shape& sh = sd->shapes[i];
      int counts[3] = {sh.num_vertices, sh.num_lines, sh.num_triangles};
      mix(counts, sizeof(counts));
      if (sh.positions && sh.num_vertices > 0) {
        const size_t n = (size_t)sh.num_vertices, take = std::min<size_t>(n, 256);
        mix(sh.positions, take * 12), mix(sh.positions + 3 * (n - take), take * 12);
      }
    }
    for (int i = 0; i < sd->num_environments; i++) mix(&sd->environments[i], offsetof(yh_environment, texels));
    ctx->scene_key = h;
  }
  ctx->d_scene_copy.reset();  // (stream_impl uploads the new table at its first launch)
  ctx->have_scene = true;
  ctx->have_state = false;
  ctx->launch_shape = 0;  // a new scene: no measured costs yet
  ctx->item_cost.clear();
  ctx->have_costs = false, ctx->costs_settled = false, ctx->dense = -1, ctx->chain = -1, ctx->chain16 = -1;
  for (double& t : ctx->shape_ms) t = 0;
  for (int& t : ctx->shape_trials) t = 0;
  ctx->trials_from_disk = false;
  lap("scene table, fingerprint");
  return YH_OK;
}

// The same tree built on the device (csrc/bvh_gpu.hip): fills `tree` like yhh::build_bvh.
int build_bvh_device(yh_context* ctx, const std::vector<yhh::Box>& boxes, yhh::Tree& tree) {
  int                n = (int)boxes.size(), num_nodes = 0, depth = 0;
  std::vector<float> nodes8((size_t)(2 * n + 1) * 8);
  tree.primitives.resize((size_t)n);
  static_assert(sizeof(yhh::Box) == 24, "boxes are passed as 6 floats");
  int e = yhk_bvh_build_gpu(n, (const float*)boxes.data(), nodes8.data(), tree.primitives.data(), &num_nodes, &depth, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "device BVH build: %s", hipGetErrorString((hipError_t)e));
  tree.nodes.resize((size_t)num_nodes);
  tree.max_depth = depth;
  for (int i = 0; i < num_nodes; i++) {
    const float* o  = &nodes8[(size_t)i * 8];
    yhh::Node&   nd = tree.nodes[(size_t)i];
    memcpy(nd.bbox.min, o, 12), memcpy(nd.bbox.max, o + 3, 12);
    int start, meta;
    memcpy(&start, o + 6, 4), memcpy(&meta, o + 7, 4);
    nd.start = start, nd.num = (short)(meta & 0xFFFF), nd.internal = (meta >> 16) & 1, nd.axis = (unsigned char)((meta >> 24) & 3);
  }
  return YH_OK;
}
