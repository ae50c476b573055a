// batch_api.cpp — the unit-level batch entry points of include/yhair.h (hair BSDF, intersection, BVH build, curves, self-tests).
#include "context_internal.h"

// ---- unit-level batches ----------------------------------------------------
namespace {
struct Staged {
  std::vector<DevBuf> bufs;
  yh_context*         ctx;
  int                 rc = YH_OK;
  explicit Staged(yh_context* c) : ctx(c) { bufs.reserve(8); }
  void* in(const void* src, size_t bytes) {
    bufs.emplace_back();
    if (rc == YH_OK) rc = upload(ctx, bufs.back(), src, bytes);
    return bufs.back().p;
  }
  void* out(size_t bytes) {
    bufs.emplace_back();
    if (rc == YH_OK) rc = alloc_zero(ctx, bufs.back(), bytes);
    return bufs.back().p;
  }
};
int finish(yh_context* ctx, int launch_err, void* dst, const void* src, size_t bytes) {
  if (launch_err) return fail(ctx, YH_E_DEVICE, "kernel launch: %s", hipGetErrorString((hipError_t)launch_err));
  YH_WAIT(ctx);
  HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return YH_OK;
}
}  // namespace

int yh_hair_brdf_batch(yh_context* ctx, int n, const yh_material* materials, const float* v, const float* normal,
    const float* tangent, float* brdf) {
  if (!ctx || n < 0 || (n && (!materials || !v || !normal || !tangent || !brdf))) return YH_E_INVALID;
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   dm = s.in(materials, sizeof(yh_material) * (size_t)n);
  auto   dv = (float*)s.in(v, 4 * (size_t)n);
  auto   dn = (float*)s.in(normal, 12 * (size_t)n);
  auto   dt = (float*)s.in(tangent, 12 * (size_t)n);
  auto   o  = (float*)s.out(120 * (size_t)n);
  if (s.rc) return s.rc;
  return finish(ctx, yhk_hair_brdf(n, dm, dv, dn, dt, o, ctx->stream), brdf, o, 120 * (size_t)n);
}
static int wowi(yh_context* ctx, int n, const float* brdf, const float* a, size_t a_floats, const float* b,
    size_t b_floats, float* out, size_t out_floats, int which) {
  if (!ctx || n < 0 || (n && (!brdf || !a || !b || !out))) return YH_E_INVALID;
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   db = (float*)s.in(brdf, 120 * (size_t)n);
  auto   da = (float*)s.in(a, 4 * a_floats * n);
  auto   dbb = (float*)s.in(b, 4 * b_floats * n);
  auto   o  = (float*)s.out(4 * out_floats * n);
  if (s.rc) return s.rc;
  int e = which == 0   ? yhk_hair_eval(n, db, da, dbb, o, ctx->stream)
          : which == 1 ? yhk_hair_sample(n, db, da, dbb, o, ctx->stream)
                       : yhk_hair_pdf(n, db, da, dbb, o, ctx->stream);
  return finish(ctx, e, out, o, 4 * out_floats * n);
}
int yh_hair_eval_batch(yh_context* ctx, int n, const float* brdf, const float* wo, const float* wi, float* f) {
  return wowi(ctx, n, brdf, wo, 3, wi, 3, f, 3, 0);
}
int yh_hair_sample_batch(yh_context* ctx, int n, const float* brdf, const float* wo, const float* rn, float* wi) {
  return wowi(ctx, n, brdf, wo, 3, rn, 2, wi, 3, 1);
}
int yh_hair_pdf_batch(yh_context* ctx, int n, const float* brdf, const float* wo, const float* wi, float* pdf) {
  return wowi(ctx, n, brdf, wo, 3, wi, 3, pdf, 1, 2);
}
int yh_hair_eval_pdf_batch(yh_context* ctx, int n, const float* brdf, const float* wo, const float* wi, float* pdf) {
  return yh_hair_pdf_batch(ctx, n, brdf, wo, wi, pdf);
}

int yh_curves_to_lines(yh_context* ctx, int n, const float* P, const float* width0, const float* width1,
    int base_vertex, float* positions, float* normals, float* radius, int* lines) {
  if (!ctx || n < 0 || (n && (!P || !width0 || !width1 || !positions || !normals || !radius || !lines))) return YH_E_INVALID;
  if (n == 0) return YH_OK;
  if (n > 400000000 || base_vertex < 0 || (long long)base_vertex + 5ll * n > 2147483647ll)
    return fail(ctx, YH_E_INVALID, "too many curves for 32-bit vertex indices");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   dp = (float*)s.in(P, 48 * (size_t)n);
  auto   d0 = (float*)s.in(width0, 4 * (size_t)n);
  auto   d1 = (float*)s.in(width1, 4 * (size_t)n);
  auto   op = (float*)s.out(60 * (size_t)n);
  auto   on = (float*)s.out(60 * (size_t)n);
  auto   orad = (float*)s.out(20 * (size_t)n);
  auto   ol = (int*)s.out(32 * (size_t)n);
  if (s.rc) return s.rc;
  int e = yhk_curves_to_lines(n, dp, d0, d1, base_vertex, op, on, orad, ol, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_curves_to_lines launch: %s", hipGetErrorString((hipError_t)e));
  YH_WAIT(ctx);
  HIPCHK(ctx, hipMemcpy(positions, op, 60 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(normals, on, 60 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(radius, orad, 20 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(lines, ol, 32 * (size_t)n, hipMemcpyDeviceToHost));
  return YH_OK;
}

int yh_bvh_build_gpu(yh_context* ctx, int n, const float* boxes, float* nodes, int* primitives) {
  if (!ctx || n < 0 || (n && !boxes)) return YH_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<yhh::Box> b((size_t)n);
  if (n) memcpy((void*)b.data(), boxes, sizeof(yhh::Box) * (size_t)n);
  yhh::Tree tree;
  int       rc = build_bvh_device(ctx, b, tree);
  if (rc) return rc;
  if (nodes)
    for (size_t i = 0; i < tree.nodes.size(); i++) {
      auto&  nd = tree.nodes[i];
      float* o  = nodes + 8 * i;
      memcpy(o, nd.bbox.min, 12), memcpy(o + 3, nd.bbox.max, 12);
      int a = nd.start, c = (int)nd.num | ((int)nd.internal << 16) | ((int)nd.axis << 24);
      memcpy(o + 6, &a, 4), memcpy(o + 7, &c, 4);
    }
  if (primitives && n) memcpy(primitives, tree.primitives.data(), sizeof(int) * (size_t)n);
  return (int)tree.nodes.size();
}

int yh_bvh_build_wide_gpu(yh_context* ctx, int n, const float* boxes, int width, float* slots) {
  if (!ctx || n < 1 || !boxes || (width != 4 && width != 8 && width != 16)) return YH_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  YH_WAIT(ctx);
  const int L = width == 4 ? 2 : width == 8 ? 3 : 4;
  DevBuf    d_boxes, d_nodes, d_pid, d_flag, d_widx, d_out;
  int       rc;
  if ((rc = upload(ctx, d_boxes, boxes, (size_t)n * 24))) return rc;
  if ((rc = alloc_zero(ctx, d_nodes, ((size_t)2 * n + 1) * 32)) || (rc = alloc_zero(ctx, d_pid, (size_t)n * 4))) return rc;
  int num_nodes = 0, levels = 0, level_first[130], count = 0;
  int e = yhk_bvh_build_resident(n, (const float*)d_boxes.p, (float*)d_nodes.p, (int*)d_pid.p, &num_nodes, &levels, level_first, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "device BVH build: %s", hipGetErrorString((hipError_t)e));
  if ((rc = alloc_zero(ctx, d_flag, ((size_t)num_nodes + 1) * 4)) || (rc = alloc_zero(ctx, d_widx, ((size_t)num_nodes + 1) * 4))) return rc;
  e = yhk_wide_index(num_nodes, (const float*)d_nodes.p, levels, level_first, L, (unsigned int*)d_flag.p, (unsigned int*)d_widx.p, &count, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "wide-node index: %s", hipGetErrorString((hipError_t)e));
  if (slots) {
    if ((rc = alloc_zero(ctx, d_out, (size_t)count * width * 32))) return rc;
    e = yhk_wide_collapse(L, num_nodes, (const float*)d_nodes.p, (const unsigned int*)d_flag.p, (const unsigned int*)d_widx.p, 1, 0, 0, d_out.p, ctx->stream);
    if (e) return fail(ctx, YH_E_DEVICE, "wide collapse: %s", hipGetErrorString((hipError_t)e));
    YH_WAIT(ctx);
    HIPCHK(ctx, hipMemcpy(slots, d_out.p, (size_t)count * width * 32, hipMemcpyDeviceToHost));
  }
  return count;
}

int yh_bvh_build_wide(int n, const float* boxes, int width, float* slots) {
  if (n < 0 || (n && !boxes) || (width != 4 && width != 8 && width != 16)) return YH_E_INVALID;
  std::vector<yhh::Box> b((size_t)n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) b[(size_t)i].min[k] = boxes[6 * (size_t)i + k], b[(size_t)i].max[k] = boxes[6 * (size_t)i + 3 + k];
  yhh::Tree tree;
  yhh::build_bvh(tree, b);
  const void* data  = nullptr;
  size_t      count = 0;
  std::vector<yhh::WideNode>   w4;
  std::vector<yhh::WideNode8>  w8;
  std::vector<yhh::WideNode16> w16;
  if (width == 4) yhh::collapse_wide(tree, w4), data = w4.data(), count = w4.size();
  if (width == 8) yhh::collapse_wide8(tree, w8), data = w8.data(), count = w8.size();
  if (width == 16) yhh::collapse_wide16(tree, w16), data = w16.data(), count = w16.size();
  if (slots && count) memcpy(slots, data, count * (size_t)width * sizeof(yhh::WideSlot));
  return (int)count;
}

int yh_bvh_build(int n, const float* boxes, float* nodes, int* primitives) {
  if (n < 0 || (n && !boxes)) return YH_E_INVALID;
  std::vector<yhh::Box> b((size_t)n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) b[(size_t)i].min[k] = boxes[6 * (size_t)i + k], b[(size_t)i].max[k] = boxes[6 * (size_t)i + 3 + k];
  yhh::Tree tree;
  yhh::build_bvh(tree, b);
  if (nodes)
    for (size_t i = 0; i < tree.nodes.size(); i++) {
      auto&  nd = tree.nodes[i];
      float* o  = nodes + 8 * i;
      memcpy(o, nd.bbox.min, 12), memcpy(o + 3, nd.bbox.max, 12);
      int a = nd.start, c = (int)nd.num | ((int)nd.internal << 16) | ((int)nd.axis << 24);
      memcpy(o + 6, &a, 4), memcpy(o + 7, &c, 4);
    }
  if (primitives && n) memcpy(primitives, tree.primitives.data(), sizeof(int) * (size_t)n);
  return (int)tree.nodes.size();
}

int yh_surface_lobe_batch(yh_context* ctx, int kind, int n, const float* params, const float* normal,
    const float* outgoing, const float* incoming, const float* rn, float* out) {
  if (!ctx || n < 0 || kind < 0 || kind >= YH_LOBE_COUNT || (n && (!params || !normal || !outgoing || !incoming || !rn || !out)))
    return YH_E_INVALID;
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   dp = (float*)s.in(params, 32 * (size_t)n);
  auto   dn = (float*)s.in(normal, 12 * (size_t)n);
  auto   da = (float*)s.in(outgoing, 12 * (size_t)n);
  auto   db = (float*)s.in(incoming, 12 * (size_t)n);
  auto   dr = (float*)s.in(rn, 12 * (size_t)n);
  auto   o  = (float*)s.out(28 * (size_t)n);
  if (s.rc) return s.rc;
  return finish(ctx, yhk_surface_lobe(kind, n, dp, dn, da, db, dr, o, ctx->stream), out, o, 28 * (size_t)n);
}
int yh_surface_bsdf_batch(yh_context* ctx, int n, const yh_material* materials, const float* normal,
    const float* outgoing, const float* incoming, const float* rn, float* out) {
  if (!ctx || n < 0 || (n && (!materials || !normal || !outgoing || !incoming || !rn || !out))) return YH_E_INVALID;
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<yhd_material> mats((size_t)n);
  for (int i = 0; i < n; i++) make_material(materials[i], mats[(size_t)i]);
  Staged s(ctx);
  auto   dm = s.in(mats.data(), sizeof(yhd_material) * (size_t)n);
  auto   dn = (float*)s.in(normal, 12 * (size_t)n);
  auto   da = (float*)s.in(outgoing, 12 * (size_t)n);
  auto   db = (float*)s.in(incoming, 12 * (size_t)n);
  auto   dr = (float*)s.in(rn, 12 * (size_t)n);
  auto   o  = (float*)s.out(4 * YH_SURFACE_BSDF_FLOATS * (size_t)n);
  if (s.rc) return s.rc;
  return finish(ctx, yhk_surface_bsdf(n, dm, dn, da, db, dr, o, ctx->stream), out, o, 4 * YH_SURFACE_BSDF_FLOATS * (size_t)n);
}

int yh_intersect_batch(yh_context* ctx, int n, const float* rays, int* object, int* element, float* uv,
    float* distance) {
  if (!ctx || n < 0 || (n && (!rays || !object || !element || !uv || !distance))) return YH_E_INVALID;
  if (!ctx->have_scene) return fail(ctx, YH_E_STATE, "yh_intersect_batch before yh_upload_scene");
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   dr = (float*)s.in(rays, 32 * (size_t)n);
  auto   dob = (int*)s.out(4 * (size_t)n);
  auto   del = (int*)s.out(4 * (size_t)n);
  auto   duv = (float*)s.out(8 * (size_t)n);
  auto   dd  = (float*)s.out(4 * (size_t)n);
  if (s.rc) return s.rc;
  // Large batches of rays that start at the reference's ray_eps (every ray the path tracer itself makes) go one lane
  // per ray through the trace-only kernel (csrc/stream.hip: k_intersect_lanes), five waves per SIMD; small
  // ones, and rays with another tmin, a quad per ray (k_intersect). Same closest hits either way.
  // YHAIR_INTERSECT=quad | lane4 | lane5 | lane6 | lane8: developer switch (waves per SIMD of the lane kernel).
  const char* mode  = getenv("YHAIR_INTERSECT");
  int         waves = 5;  // 91 registers without a spill: five waves per SIMD (6 and 8 spill 39 / 61 registers, measured slower)
  bool        lanes = n >= 65536 && lane_kernels_can_address(ctx);
  if (mode && !strcmp(mode, "quad")) lanes = false;
  else if (mode && !strncmp(mode, "lane", 4)) {
    if (!lane_kernels_can_address(ctx))  // (beyond 4 GB the buffer resource clamps and the loads return zeros: wrong hits, no error)
      return fail(ctx, YH_E_INVALID, "YHAIR_INTERSECT=%s: the one-lane kernel reads the scene's trees through 32-bit byte offsets and this scene's exceed 4 GB", mode);
    lanes = true, waves = std::max(4, std::min(8, atoi(mode + 4)));
  }
  for (int i = 0; lanes && i < n; i++) lanes = rays[8 * (size_t)i + 6] == 1e-4f;
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if (lanes) {
    const int occupancy = std::min(waves, yhk_intersect_lanes_occupancy(&ctx->scene, waves));  // (256-thread blocks: one wave per SIMD each)
    if (occupancy < 1) return fail(ctx, YH_E_DEVICE, "k_intersect_lanes cannot run with its LDS layout on this device");
    const int    grid        = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)n + 255) / 256, (int64_t)ctx->num_cus * occupancy));
    const int    ovf_entries = 2 * std::max(8, ctx->stack_need);
    auto         dcur        = (int*)s.out(16);
    auto         dovf        = (unsigned int*)s.out((size_t)grid * 4 * ovf_entries * 64 * 4);
    if (s.rc) return s.rc;
    if (!ctx->d_scene_copy.p) {
      int rc = upload(ctx, ctx->d_scene_copy, &ctx->scene, sizeof(yhd_scene));
      if (rc) return rc;
    }
    int e = yhk_intersect_lanes(&ctx->scene, (const yhd_scene*)ctx->d_scene_copy.p, n, dr, dcur, dovf, ovf_entries, dob, del, duv, dd, waves, grid, ctx->stream);
    if (e) return fail(ctx, YH_E_DEVICE, "k_intersect_lanes launch: %s", hipGetErrorString((hipError_t)e));
  } else {
    int e = yhk_intersect(&ctx->scene, n, dr, dob, del, duv, dd, ctx->stream);
    if (e) return fail(ctx, YH_E_DEVICE, "k_intersect launch: %s", hipGetErrorString((hipError_t)e));
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  YH_WAIT(ctx);
  HIPCHK(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->ev0, ctx->ev1));  // (yh_last_trace_ms: the kernel alone, without the copies)
  ctx->last_launches = 1;
  HIPCHK(ctx, hipMemcpy(object, dob, 4 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(element, del, 4 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(uv, duv, 8 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(distance, dd, 4 * (size_t)n, hipMemcpyDeviceToHost));
  return YH_OK;
}

// ---- the four self-tests (ext.cpp:555-693) ---------------------------------
// The host replays the reference's serial structure (seed, loop bounds with
// the accumulating float counters, per-block draw counts) and hands every
// (beta_m, beta_n) block to the device with the generator state at its start.
int yh_selftest(yh_context* ctx, int which, float* worst) {
  if (!ctx) return YH_E_INVALID;
  if (which < 0 || which > 3) return fail(ctx, YH_E_INVALID, "unknown self-test %d", which);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf sums, wbits;
  int    rc;
  if ((rc = alloc_zero(ctx, sums, 6 * sizeof(double)))) return rc;
  if ((rc = alloc_zero(ctx, wbits, 4))) return rc;
  auto lum = [](const double* s) { return (float)(0.2126 * s[0] + 0.7152 * s[1] + 0.0722 * s[2]); };
  auto sample_sphere = [](float rx, float ry, float* w) {  // math.h:4847-4852
    float z = 2 * ry - 1;
    float r = std::sqrt(fmin_(fmax_(1 - z * z, 0.0f), 1.0f));
    float phi = 2 * pif * rx;
    w[0] = r * std::cos(phi), w[1] = r * std::sin(phi), w[2] = z;
  };
  Rng   rng = make_rng(199382389514ULL);
  float wo[3] = {0, 0, 1};
  if (which == 0 || which == 1) {
    float x = rand1f(rng), y = rand1f(rng);
    sample_sphere(x, y, wo);
  }
  bool  ok  = true;
  float dev = 0;
  auto run = [&](float bm, float bn, int count, int per_iter, double* out, float* dmax) -> int {
    HIPCHK(ctx, hipMemsetAsync(sums.p, 0, 6 * sizeof(double), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(wbits.p, 0, 4, ctx->stream));
    int e = yhk_selftest(which, bm, bn, rng.state, rng.inc, count, wo, (double*)sums.p, (unsigned int*)wbits.p, ctx->stream);
    if (e) return fail(ctx, YH_E_DEVICE, "k_selftest launch: %s", hipGetErrorString((hipError_t)e));
    YH_WAIT(ctx);
    HIPCHK(ctx, hipMemcpy(out, sums.p, 6 * sizeof(double), hipMemcpyDeviceToHost));
    unsigned int bits;
    HIPCHK(ctx, hipMemcpy(&bits, wbits.p, 4, hipMemcpyDeviceToHost));
    memcpy(dmax, &bits, 4);
    skip_rng(rng, (uint64_t)count * per_iter);
    return YH_OK;
  };
  double s[6];
  float  d;
  if (which == 0 || which == 1) {
    for (float bm = 0.1f; bm < 1.0f; bm += 0.2f)
      for (float bn = 0.1f; bn < 1.0f; bn += 0.2f) {
        const int count = 300000;
        if ((rc = run(bm, bn, count, 3, s, &d))) return rc;
        float avg = which == 0 ? lum(s) / (count * (1 / (4 * pif))) : lum(s) / count;
        float lo = which == 0 ? 0.95f : 0.99f, hi = which == 0 ? 1.05f : 1.01f;
        if (!(avg >= lo && avg <= hi)) ok = false;
        dev = fmax_(dev, std::fabs(avg - 1));
      }
  } else if (which == 2) {
    for (float bm = 0.1f; bm < 1.0f; bm += 0.2f)
      for (float bn = 0.4f; bn < 1.0f; bn += 0.2f) {
        if ((rc = run(bm, bn, 10000, 5, s, &d))) return rc;
        if (!(d <= 0.001f)) ok = false;
        dev = fmax_(dev, d);
      }
  } else {
    for (float bm = 0.2f; bm < 1.0f; bm += 0.2f)
      for (float bn = 0.4f; bn < 1.0f; bn += 0.2f) {
        const int count = 64 * 1024;
        float x = rand1f(rng), y = rand1f(rng);
        sample_sphere(x, y, wo);
        if ((rc = run(bm, bn, count, 3, s, &d))) return rc;
        float fi = lum(s) / count, fu = lum(s + 3) / (count * (1 / (4 * pif)));
        float err = std::fabs(fi - fu) / fu;
        if (err >= 0.05f) ok = false;
        dev = fmax_(dev, err);
      }
  }
  if (worst) *worst = dev;
  if (!ok) return fail(ctx, YH_E_SELFTEST, "TEST FAILED! (self-test %d, worst deviation %g)", which, dev);
  return YH_OK;
}
