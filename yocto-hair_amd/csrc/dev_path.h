// dev_path.h — shading points, lights and the one-sample-MIS path integrator.
//
// Restates pt.cpp:211-229 (camera), 232-369 (shading point), 405-492
// (eval_brdf: only the diffuse and hair lobes are reachable in scope),
// 536-547 + 148-200 (environment lookup), 1069-1256 (lobe dispatch),
// 1283-1358 (light sampling and its pdf), 1380-1511 (trace_path) and
// 1676-1689 (trace_sample).
//
// The RNG draw order is the one the g++-built reference executes (SURVEY.md
// 3.2): lens uv before pixel uv; `rn` (x, y) before `rnl`; `ruv` (x, y), `rel`,
// `rl` for light sampling.
#ifndef YH_DEV_PATH_H_
#define YH_DEV_PATH_H_
#include "dev_hair.h"
#include "dev_surface.h"
#include "dev_trace.h"
#if YH_LANE
#include "dev_lane.h"
#endif

namespace yhd {

// pt.cpp:211-229 with sample_disk (math.h:4895-4899)
YH_DEV ray_t sample_camera(const yhd_camera& cam, int i, int j, int w, int h, float pu, float pv,
    float lu, float lv) {
  // (called by the four lanes of a quad with identical arguments: divisions spread over the lanes)
  const unsigned int ql = __lane_id() & 3u;
  float uvq = (ql == 0 ? (float)i + pu : (float)j + pv) / (ql == 0 ? (float)w : (float)h);
  float uvx = quad_bcast_f<0>(uvq), uvy = quad_bcast_f<1>(uvq);
  f3    q  = {cam.film_x * (0.5f - uvx), cam.film_y * (uvy - 0.5f), cam.lens};
  f3    dc = -quad_normalize(q);
  f3    e  = mk3(0.0f);
  if (cam.aperture != 0) {  // pinhole: lens point = finite * 0 / 2 = 0 whatever sample_disk returns
    float r   = sqrtf(lv);
    float phi = 2 * pif * lu;
    float lx = cosf(phi) * r, ly = sinf(phi) * r;
    e = f3{lx * cam.aperture / 2, ly * cam.aperture / 2, 0};
  }
  f3    p  = quad_div(dc * cam.focus, fabs_(dc.z));
  f3    d  = quad_normalize(p - e);
  frame f  = ldframe(cam.frame);
  return mkray(transform_point(f, e), quad_normalize(transform_vector(f, d)));
}

// The same by ONE lane (the staged integrators, wavefront.hip / stream.hip): the expressions of the quad
// form above, which only spreads their divisions over four lanes, so the same bits.
YH_DEV ray_t sample_camera_lane(const yhd_camera& cam, int i, int j, int w, int h, float pu, float pv, float lu, float lv) {
  float uvx = ((float)i + pu) / (float)w, uvy = ((float)j + pv) / (float)h;
  f3    q   = {cam.film_x * (0.5f - uvx), cam.film_y * (uvy - 0.5f), cam.lens};
  f3    dc  = -normalize(q);
  f3    e   = mk3(0.0f);
  if (cam.aperture != 0) {
    float r   = sqrtf(lv);
    float phi = 2 * pif * lu;
    float lx = cosf(phi) * r, ly = sinf(phi) * r;
    e = f3{lx * cam.aperture / 2, ly * cam.aperture / 2, 0};
  }
  f3    p = (dc * cam.focus) / fabs_(dc.z);
  f3    d = normalize(p - e);
  frame f = ldframe(cam.frame);
  return mkray(transform_point(f, e), normalize(transform_vector(f, d)));
}

YH_DEV f3 transform_normal(const frame& a, f3 b) { return normalize(transform_vector(a, b)); }
YH_DEV f3 quad_transform_normal(const frame& a, f3 b) { return quad_normalize(transform_vector(a, b)); }

// eval_position (pt.cpp:232-250) of a point on ELEMENT `element` (used to
// sample area lights by triangle index): goes through elems / vpos.
YH_DEV f3 eval_position(const yhd_scene& sc, const yhd_object& o, int element, float u, float v) {
  frame    fr = ldframe(o.frame);
  yhd_int4 e  = sc.elems[o.elem_base + element];
  if (o.kind == YH_KIND_TRIANGLES) {
    f3 p0 = xyz(sc.vpos[o.vert_base + e.x]), p1 = xyz(sc.vpos[o.vert_base + e.y]),
       p2 = xyz(sc.vpos[o.vert_base + e.z]);
    return transform_point(fr, p0 * (1 - u - v) + p1 * u + p2 * v);
  } else {
    f3 p0 = xyz(sc.vpos[o.vert_base + e.x]), p1 = xyz(sc.vpos[o.vert_base + e.y]);
    return transform_point(fr, p0 * (1 - u) + p1 * u);
  }
}

// Geometry of a HIT, from its leaf record (one fetch, no index chasing):
// eval_position (pt.cpp:232-250), eval_normal (272-292) with its
// eval_element_normal fallback (253-269).
struct hit_geom {
  f3 position, normal, element_normal;
};
// (quad-uniform: the normalisations use the quad forms of dev_math.h)
YH_DEV hit_geom eval_hit(const yhd_scene& sc, const yhd_object& o, int slot, float u, float v) {
  frame    fr = ldframe(o.frame);
  hit_geom g;
  if (o.kind == YH_KIND_TRIANGLES) {
    const yhd_float4* rec = sc.prims + (size_t)o.prim_base + (size_t)slot * 6;
    f3 p0 = xyz(ldg4(rec)), p1 = xyz(ldg4(rec + 1)), p2 = xyz(ldg4(rec + 2));
    g.position       = transform_point(fr, p0 * (1 - u - v) + p1 * u + p2 * v);
    g.element_normal = quad_transform_normal(fr, quad_normalize(cross(p1 - p0, p2 - p0)));
    if (o.has_normals) {
      f3 n0 = xyz(ldg4(rec + 3)), n1 = xyz(ldg4(rec + 4)), n2 = xyz(ldg4(rec + 5));
      g.normal = quad_transform_normal(fr, quad_normalize(n0 * (1 - u - v) + n1 * u + n2 * v));
    } else {
      g.normal = g.element_normal;
    }
  } else {
    const yhd_float4* rec = sc.prims + (size_t)o.prim_base + (size_t)slot * 4;
    f3 p0 = xyz(ldg4(rec)), p1 = xyz(ldg4(rec + 1));
    g.position       = transform_point(fr, p0 * (1 - u) + p1 * u);
    g.element_normal = quad_transform_normal(fr, quad_normalize(p1 - p0));
    if (o.has_normals) {
      f3 n0 = xyz(ldg4(rec + 2)), n1 = xyz(ldg4(rec + 3));
      g.normal = quad_transform_normal(fr, quad_normalize(n0 * (1 - u) + n1 * u));
    } else {
      g.normal = g.element_normal;
    }
  }
  return g;
}

// texture lookup of an environment (pt.cpp:167-200), wrap + bilinear
YH_DEV f3 eval_env_texture(const yhd_scene& sc, const yhd_environment& env, float u_, float v_) {
  if (env.tex_w == 0) return mk3(1.0f);
  int   sx = env.tex_w, sy = env.tex_h;
  float s = fmodf(u_, 1.0f) * sx;
  if (s < 0) s += sx;
  float t = fmodf(v_, 1.0f) * sy;
  if (t < 0) t += sy;
  int   i = iclamp((int)s, 0, sx - 1), j = iclamp((int)t, 0, sy - 1);
  int   ii = (i + 1) % sx, jj = (j + 1) % sy;
  float u = s - i, v = t - j;
  const yhd_float4* tx = sc.env_texels + env.texel_base;
  f3 a = xyz(tx[(size_t)j * sx + i]), b = xyz(tx[(size_t)jj * sx + i]);
  f3 c = xyz(tx[(size_t)j * sx + ii]), d = xyz(tx[(size_t)jj * sx + ii]);
  return a * (1 - u) * (1 - v) + b * (1 - u) * v + c * u * (1 - v) + d * u * v;
}
// eval_environment (pt.cpp:536-547)
template <bool COUNT>
YH_DEV f3 eval_environment(const trace_ctx& tc, f3 dir) {
  const yhd_scene& sc = *tc.sc;
  f3 emission = mk3(0.0f);
  for (int k = 0; k < sc.num_environments; k++) {
    const yhd_environment& env = sc.environments[k];
    if (COUNT) count_quad<COUNT>(tc.stats->envl);
    if (env.tex_w == 0) {  // no texture: eval_texture returns {1,1,1} whatever the coordinates are
      emission = emission + ld3(env.emission) * mk3(1.0f);
      continue;
    }
    f3    wl = transform_direction(ldframe(env.inv_frame), dir);
    float tx = atan2f(wl.z, wl.x) / (2 * pif);
    float ty = acosf(fclamp(wl.y, -1.0f, 1.0f)) / pif;
    if (tx < 0) tx += 1;
    emission = emission + ld3(env.emission) * eval_env_texture(sc, env, tx, ty);
  }
  return emission;
}

// std::upper_bound (math.h:4960): first index in [lo, lo + len) whose entry is greater than r, lo + len if none
template <typename P>
YH_DEV int cdf_upper_bound(P cdf, int lo, int len, float r) {
  while (len > 0) {
    int half = len >> 1;
    if (!(r < cdf[lo + half])) {
      lo += half + 1;
      len -= half + 1;
    } else {
      len = half;
    }
  }
  return lo;
}
// sample_discrete_cdf (math.h:4957-4962): std::upper_bound over the cdf
template <typename P>
YH_DEV int sample_discrete_cdf(P cdf, int n, float r) {
  float back = cdf[n - 1];
  r          = fclamp(r * back, 0.0f, back - 0.00001f);
  return iclamp(cdf_upper_bound(cdf, 0, n, r), 0, n - 1);
}
// The same through a coarse index in LDS (yhd_scene::env_tab: tab[k] = cdf[min(n, (k + 1) * S) - 1], K entries): the
// block of S entries that holds the answer is found in LDS, only the search inside it reads memory. Entries before
// the block are <= r and the block's last entry is > r, so the result is upper_bound's over the whole array.
YH_DEV int sample_discrete_cdf_indexed(const float* cdf, int n, const YH_LDS float* tab, int K, int S, float r) {
  float back = tab[K - 1];  // = cdf[n - 1]
  r          = fclamp(r * back, 0.0f, back - 0.00001f);
  int k      = cdf_upper_bound(tab, 0, K, r);
  if (k >= K) return n - 1;  // no entry is greater than r: upper_bound = n, clamped
  int lo = k * S;
  return iclamp(cdf_upper_bound(cdf, lo, min(S, n - lo), r), 0, n - 1);
}

// ---- small area lights (yh_device.h: YH_SMALL_LIGHT_F4 record in LDS) ----------------------------------------
// object -> world frame and its inverse, from the LDS copy of the object table when there is one
YH_DEV void object_frames(const trace_ctx& tc, int object, frame& fr, frame& inv) {
  if (tc.lds_scene) {
    const YH_LDS v4f* ob = tc.lds_scene + YH_OBJECT_F4 * object;
    v4f a = ob[0], b = ob[1], c = ob[2], d = ob[3], e = ob[4], g = ob[5];
    fr.x = {a.x, a.y, a.z}, fr.y = {a.w, b.x, b.y}, fr.z = {b.z, b.w, c.x}, fr.o = {c.y, c.z, c.w};
    inv.x = {d.x, d.y, d.z}, inv.y = {d.w, e.x, e.y}, inv.z = {e.z, e.w, g.x}, inv.o = {g.y, g.z, g.w};
  } else {
    const yhd_object& o = tc.sc->objects[object];
    fr = ldframe(o.frame), inv = ldframe(o.inv_frame);
  }
}
// Closest hit of a ray with a small light: what intersect_instance_bvh (pt.cpp:1031-1037) computes for a shape
// whose tree is one leaf — the ray taken to object space (pt.cpp:1012-1013), the root box, then the leaf's
// triangles in order with tmax shrinking after each accepted hit (pt.cpp:905-923) — on the record in LDS: the
// same arithmetic as the traversal (dev_trace.h), no memory access.
YH_DEV bool small_light_hit(const YH_LDS v4f* L, const frame& inv, f3 ro, f3 rd, int& slot, float& uu, float& vv) {
  f3  lo = transform_point(inv, ro), ld = transform_vector(inv, rd);
  f3  ldinv = quad_rcp(ld);
  v4f b0 = L[0], b1 = L[1];
  if (!intersect_bbox(lo, ldinv, ray_eps, flt_max, xyz(b0), xyz(b1))) return false;
  const int num  = __float_as_int(b0.w);
  float     tmax = flt_max;
  bool      hit  = false;
  for (int i = 0; i < num; i++) {
    float u = 0, v = 0, d = 0;
    if (intersect_triangle(lo, ld, ray_eps, tmax, xyz(L[2 + 3 * i]), xyz(L[3 + 3 * i]), xyz(L[4 + 3 * i]), u, v, d)) hit = true, slot = i, uu = u, vv = v, tmax = d;
  }
  return hit;
}

// sample_lights (pt.cpp:1283-1308)
// BIG_LIGHTS = false: every area light of the scene is a small one with its record in LDS (the host says so:
// yhd_scene::general_materials is set otherwise), and the code that samples / intersects a light through memory —
// with a whole traversal loop inlined into the shading code — is not compiled in.
template <bool COUNT, bool BIG_LIGHTS>
YH_DEV f3 sample_lights(const trace_ctx& tc, f3 position, float rl, float rel, float ruvx,
    float ruvy) {
  const yhd_scene& sc = *tc.sc;
  int n        = sc.num_lights;
  int light_id = iclamp((int)(rl * n), 0, n - 1);
  const yhd_light& light = sc.lights[light_id];
  if (light.object >= 0 && (!BIG_LIGHTS || (light.small_base >= 0 && tc.lds_lights))) {  // a small light: its record is in LDS
    const YH_LDS v4f* L = tc.lds_lights + light.small_base;
    int   element = sample_discrete_cdf((const YH_LDS float*)(L + 14), light.cdf_count, rel);
    int   rec     = 0;
    for (int i = 1; i < __float_as_int(L[0].w); i++)
      if (__float_as_int(L[2 + 3 * i].w) == element) rec = i;
    float su = sqrtf(ruvx);
    float u = 1 - su, v = ruvy * su;  // sample_triangle, math.h:4910-4912
    frame fr, inv;
    object_frames(tc, light.object, fr, inv);
    f3 p0 = xyz(L[2 + 3 * rec]), p1 = xyz(L[3 + 3 * rec]), p2 = xyz(L[4 + 3 * rec]);
    f3 lposition = transform_point(fr, p0 * (1 - u - v) + p1 * u + p2 * v);  // eval_position (pt.cpp:232-250)
    return normalize(lposition - position);
  } else if (BIG_LIGHTS && light.object >= 0) {
    int   element = sample_discrete_cdf(sc.light_cdf + light.cdf_base, light.cdf_count, rel);
    float su      = sqrtf(ruvx);
    float u = 1 - su, v = ruvy * su;  // sample_triangle, math.h:4910-4912
    f3 lposition = eval_position(sc, sc.objects[light.object], element, u, v);
    return normalize(lposition - position);
  } else if (light.environment >= 0) {
    const yhd_environment& env = sc.environments[light.environment];
    if (env.tex_w) {
      if (COUNT) count_quad<COUNT>(tc.stats->envs);
      int   idx = (tc.lds_envtab && light_id == sc.env_tab_light)
                      ? sample_discrete_cdf_indexed(sc.light_cdf + light.cdf_base, light.cdf_count, tc.lds_envtab, sc.env_tab_k, sc.env_tab_stride, rel)
                      : sample_discrete_cdf(sc.light_cdf + light.cdf_base, light.cdf_count, rel);
      float ux  = (idx % env.tex_w + 0.5f) / env.tex_w;
      float uy  = (idx / env.tex_w + 0.5f) / env.tex_h;
      return transform_direction(ldframe(env.frame),
          f3{cosf(ux * 2 * pif) * sinf(uy * pif), cosf(uy * pif), sinf(ux * 2 * pif) * sinf(uy * pif)});
    } else {
      // sample_sphere (math.h:4847-4852)
      float z   = 2 * ruvy - 1;
      float r   = sqrtf(fclamp(1 - z * z, 0.0f, 1.0f));
      float phi = 2 * pif * ruvx;
      return f3{r * cosf(phi), r * sinf(phi), z};
    }
  }
  return mk3(0.0f);
}

// sample_lights_pdf (pt.cpp:1311-1358)
template <bool COUNT, int STRIDE, bool BIG_LIGHTS>
YH_DEV float sample_lights_pdf(const trace_ctx& tc, f3 position, f3 direction) {
  const yhd_scene& sc = *tc.sc;
  float pdf = 0.0f;
  for (int k = 0; k < sc.num_lights; k++) {
    const yhd_light& light = sc.lights[k];
    if (light.object >= 0 && (!BIG_LIGHTS || (light.small_base >= 0 && tc.lds_lights))) {  // a small light: no memory access
      const YH_LDS v4f* L = tc.lds_lights + light.small_base;
      frame fr, inv;
      object_frames(tc, light.object, fr, inv);
      const float area = L[1].w;
      float lpdf = 0.0f;
      f3    next_position = position;
      for (int bounce = 0; bounce < 100; bounce++) {
        int   slot = 0;
        float uu = 0, vv = 0;
        if (!small_light_hit(L, inv, next_position, direction, slot, uu, vv)) break;
        f3 p0 = xyz(L[2 + 3 * slot]), p1 = xyz(L[3 + 3 * slot]), p2 = xyz(L[4 + 3 * slot]);
        f3 lposition = transform_point(fr, p0 * (1 - uu - vv) + p1 * uu + p2 * vv);              // eval_hit: position
        f3 lnormal   = quad_transform_normal(fr, quad_normalize(cross(p1 - p0, p2 - p0)));        // eval_hit: element normal
        f3 dp        = lposition - position;
        lpdf += dot(dp, dp) / (fabs_(dot(lnormal, direction)) * area);
        next_position = lposition + direction * 1e-3f;
      }
      pdf += lpdf;
    } else if (BIG_LIGHTS && light.object >= 0) {
      const yhd_object& o = sc.objects[light.object];
      float lpdf = 0.0f;
      f3    next_position = position;
      for (int bounce = 0; bounce < 100; bounce++) {
        hit_t isec;
#if YH_LANE
        isec = lane_trace(tc, *tc.ls, next_position, direction, light.object);
#else
        isec = trace_ray<COUNT, STRIDE>(tc, mkray(next_position, direction), light.object);
#endif
        if (isec.object < 0) break;
        hit_geom lg        = eval_hit(sc, o, isec.slot, isec.u, isec.v);
        f3       lposition = lg.position;
        f3       lnormal   = lg.element_normal;
        float area      = sc.light_cdf[light.cdf_base + light.cdf_count - 1];
        f3    dp        = lposition - position;
        lpdf += dot(dp, dp) / (fabs_(dot(lnormal, direction)) * area);
        next_position = lposition + direction * 1e-3f;
      }
      pdf += lpdf;
    } else if (light.environment >= 0) {
      const yhd_environment& env = sc.environments[light.environment];
      if (env.tex_w) {
        f3    wl = transform_direction(ldframe(env.inv_frame), direction);
        float tx = atan2f(wl.z, wl.x) / (2 * pif);
        float ty = acosf(fclamp(wl.y, -1.0f, 1.0f)) / pif;
        if (tx < 0) tx += 1;
        int   i   = iclamp((int)(tx * env.tex_w), 0, env.tex_w - 1);
        int   j   = iclamp((int)(ty * env.tex_h), 0, env.tex_h - 1);
        int   idx = j * env.tex_w + i;
        const float* cdf = sc.light_cdf + light.cdf_base;
        float prob = (idx == 0 ? cdf[0] : cdf[idx] - cdf[idx - 1]) / cdf[light.cdf_count - 1];
        float angle = (2 * pif / env.tex_w) * (pif / env.tex_h) * sinf(pif * (j + 0.5f) / env.tex_h);
        pdf += prob / angle;
      } else {
        pdf += 1 / (4 * pif);
      }
    }
  }
  pdf *= (float)1 / (float)sc.num_lights;
  return pdf;
}

#if YH_LANE
// sample_lights_pdf OUT OF LINE for k_stream<GENERAL> (one lane per path, big lights possible): inlined — twice, path_step calls it from its volume
// and its surface branch — the lights' loop with a whole traversal loop inside (the pdf walk of a big light: up to a hundred
// intersect_instance_bvh rays, pt.cpp:1315-1334) was what pushed that kernel's shading stage into scratch: 278 spilled registers and scratch
// instructions inside three of its loops. As a function of its own it has its own register budget; the caller pays one
// call per bounce. Everything by value, as lane_trace_exact: the caller's stack state stays in registers.
struct lane_lights_pdf_result {
  float pdf;
  int   base;  // the stack's window base afterwards (sp is back where it was)
};
__device__ __attribute__((noinline)) lane_lights_pdf_result lane_lights_pdf_general(const yhd_scene* sc, const YH_LDS v4f* lds_scene, const YH_LDS v4f* lds_lights,
    const YH_LDS float* lds_envtab, YH_LDS unsigned int* lds, unsigned int* ovf, int sp, int base, f3 position, f3 direction) {
  trace_ctx tc;
  tc.sc = sc, tc.sc_dev = sc, tc.lds_stack = nullptr, tc.lds_scene = lds_scene, tc.stats = nullptr;
  tc.lds_lights = lds_lights, tc.lds_envtab = lds_envtab, tc.lds_mats = nullptr;
  lane_stack s;
  s.lds = lds, s.ovf = ovf, s.sp = sp, s.base = base;
  tc.ls = &s;
  const float pdf = sample_lights_pdf<false, 64, true>(tc, position, direction);
  return lane_lights_pdf_result{pdf, s.base};
}
template <bool GENERAL>
YH_DEV float lane_lights_pdf(const trace_ctx& tc, f3 position, f3 direction) {
  if constexpr (GENERAL) {
    const lane_lights_pdf_result r = lane_lights_pdf_general(tc.sc_dev, tc.lds_scene, tc.lds_lights, tc.lds_envtab, tc.ls->lds, tc.ls->ovf, tc.ls->sp, tc.ls->base, position, direction);
    tc.ls->base = r.base;
    return r.pdf;
  } else {
    return sample_lights_pdf<false, 64, false>(tc, position, direction);
  }
}
#endif

// State of one path in flight (the locals of trace_path, pt.cpp:1383-1387).
struct path_t {
  ray_t ray;
  f3    radiance, weight;
  int   bounce;
  bool  hit;
  // the reference's volume_stack (pt.cpp:1385): it only pushes when empty and pops
  // otherwise, so it never holds more than one medium. Used by GENERAL kernels only.
  bool   in_medium;
  vsdf_t medium;
#if YH_LANE
  // one lane per path (k_stream): the medium stays in the path's slot — path_step reads it where the free flight needs it and medium_crossing
  // writes it when the path enters one — instead of travelling through the whole shading code in seven registers (dead in the hair stage)
  yhd_float4* medium_mem;
#endif
};

// End of one bounce (pt.cpp:1499-1507): weight check, Russian roulette, bounce count.
YH_DEV bool path_continue(path_t& ps, rng_t& rng, int bounces) {
  if (is_zero(ps.weight) || !finite3(ps.weight)) return false;
  if (ps.bounce > 3) {
    float rr_prob = fmin_(0.99f, hmax(ps.weight));
    if (rand1f(rng) >= rr_prob) return false;
    ps.weight = ps.weight * (1 / rr_prob);
  }
  ps.bounce++;
  return ps.bounce < bounces;
}

// eval_texcoord (pt.cpp:295-311) of a hit: the element's vertex texture coordinates interpolated
// like positions (math.h:3322-3331), or the element uv when the shape has none.
YH_DEV void eval_texcoord(const yhd_scene& sc, const yhd_object& o, const hit_t& isec, float& tu, float& tv) {
  tu = isec.u, tv = isec.v;
  if (!o.has_texcoords) return;
  const bool lines   = o.kind == YH_KIND_LINES;
  int        element = __float_as_int(lines ? sc.prims[(size_t)o.prim_base + (size_t)isec.slot * 4 + 2].w
                                            : sc.prims[(size_t)o.prim_base + (size_t)isec.slot * 6].w);
  yhd_int4   e       = sc.elems[o.elem_base + element];
  const float* t0 = sc.vtex + 2 * (size_t)(o.vert_base + e.x);
  const float* t1 = sc.vtex + 2 * (size_t)(o.vert_base + e.y);
  if (lines) {
    tu = t0[0] * (1 - isec.u) + t1[0] * isec.u, tv = t0[1] * (1 - isec.u) + t1[1] * isec.u;
  } else {
    const float* t2 = sc.vtex + 2 * (size_t)(o.vert_base + e.z);
    tu = t0[0] * (1 - isec.u - isec.v) + t1[0] * isec.u + t2[0] * isec.v;
    tv = t0[1] * (1 - isec.u - isec.v) + t1[1] * isec.u + t2[1] * isec.v;
  }
}

// Entering / leaving a closed transmissive object (pt.cpp:1458-1467); the medium entered is
// eval_vsdf at the crossing point (pt.cpp:504-527).
YH_DEV void medium_crossing(const yhd_scene& sc, path_t& ps, const yhd_material& mat, f3 normal, f3 outgoing,
    f3 incoming, f3 color_tex, float emission_tex_x, float tu, float tv) {
  if (!mat.has_volume || !(dot(normal, outgoing) * dot(normal, incoming) < 0)) return;
  if (!ps.in_medium) {
    if (mat.color_tex >= 0 || mat.emission_tex >= 0) {
      f3    base         = ld3(mat.color) * color_tex;
      float transmission = mat.transmission * emission_tex_x;
      ps.medium.density  = mk3(0.0f);
      if (transmission) {  // thin is false here (has_volume)
        f3 c = {fclamp(base.x, 0.0001f, 1.0f), fclamp(base.y, 0.0001f, 1.0f), fclamp(base.z, 0.0001f, 1.0f)};
        ps.medium.density = -f3{logf(c.x), logf(c.y), logf(c.z)} / mat.trdepth;
      }
    } else {
      ps.medium.density = ld3(mat.vol_density);  // the same expression, evaluated on the host
    }
    ps.medium.scatter    = ld3(mat.vol_scatter) * eval_texture(sc, mat.scattering_tex, false, tu, tv);
    ps.medium.anisotropy = mat.vol_anisotropy;
#if YH_LANE
    ps.medium_mem[0] = yhd_float4{ps.medium.density.x, ps.medium.density.y, ps.medium.density.z, ps.medium.anisotropy};
    ps.medium_mem[1] = yhd_float4{ps.medium.scatter.x, ps.medium.scatter.y, ps.medium.scatter.z, 0.0f};
#endif
  }
  ps.in_medium = !ps.in_medium;
}

// One iteration of trace_path's bounce loop (pt.cpp:1395-1508) given the
// closest hit of ps.ray. Returns true when the path continues with the new
// ps.ray, false when it ended (miss, zero / non-finite weight, Russian
// roulette or the bounce limit).
template <bool COUNT, int STRIDE, bool GENERAL>
YH_DEV bool path_step(const trace_ctx& tc, path_t& ps, const hit_t& isec, rng_t& rng, int bounces) {
  const yhd_scene& sc = *tc.sc;
  unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0;
  if (COUNT) k0 = clock64();
#if YH_LANE
  if (isec.object < 0) return false;  // (k_stream's shading stages take hits only: a ray that missed goes to its finish stage, which looks the environment up)
#else
  if (isec.object < 0) {
    ps.radiance = ps.radiance + ps.weight * eval_environment<COUNT>(tc, ps.ray.d);
    if (COUNT) tc.stats->c_rest += clock64() - k0;
    return false;
  }
#endif
  f3 outgoing = -ps.ray.d;
  if (GENERAL && ps.in_medium) {
#if YH_LANE
    {
      const yhd_float4 m0 = ps.medium_mem[0], m1 = ps.medium_mem[1];
      ps.medium.density = f3{m0.x, m0.y, m0.z}, ps.medium.anisotropy = m0.w, ps.medium.scatter = f3{m1.x, m1.y, m1.z};
    }
#endif
    // free flight inside the medium (pt.cpp:1403-1414); g++ draws rd before rl
    float rd = rand1f(rng), rl = rand1f(rng);
    float dist = sample_transmittance(ps.medium.density, isec.distance, rl, rd);
    ps.weight  = ps.weight * (eval_transmittance(ps.medium.density, dist) /
                                sample_transmittance_pdf(ps.medium.density, dist, isec.distance));
    if (dist < isec.distance) {  // scatter before reaching the surface (pt.cpp:1472-1497)
      f3 position = ps.ray.o + ps.ray.d * dist;
      ps.hit      = true;
      f3 incoming;
      if (rand1f(rng) < 0.5f) {
        float rnx = rand1f(rng), rny = rand1f(rng);
        float rnl = rand1f(rng);
        (void)rnl;
        incoming = sample_scattering(ps.medium, outgoing, rnx, rny);
      } else {
        float ruvx = rand1f(rng), ruvy = rand1f(rng);
        float rel = rand1f(rng);
        float rl2 = rand1f(rng);
        incoming  = sample_lights<COUNT, GENERAL>(tc, position, rl2, rel, ruvx, ruvy);
      }
      f3    f         = eval_scattering(ps.medium, outgoing, incoming);
      float pdf       = sample_scattering_pdf(ps.medium, outgoing, incoming);
#if YH_LANE
      float light_pdf = lane_lights_pdf<GENERAL>(tc, position, incoming);
#else
      float light_pdf = sample_lights_pdf<COUNT, STRIDE, GENERAL>(tc, position, incoming);
#endif
      ps.weight = ps.weight * (f / (0.5f * pdf + 0.5f * light_pdf));
      ps.ray    = mkray(position, incoming);
      if (COUNT) tc.stats->c_rest += clock64() - k0;
      return path_continue(ps, rng, bounces);
    }
  }
  // The object and material records of the hit: the plain kernel variants (GENERAL = false: the host guarantees
  // that the scene-level table and the material table are staged in LDS) read them there, so that shading a hit
  // starts with ONE dependent fetch (the leaf record) instead of object -> record and object -> material chains.
  // (not in the 256-thread k_trace shape, STRIDE 64 quads: there the LDS form happens to put spill reloads into
  // the traversal loop, 0.8x on C2 / C4 — tools/check_codegen.py watches for that; k_stream is YH_LANE)
  constexpr bool from_lds = !GENERAL && (YH_LANE || STRIDE != 64);
  const yhd_object&   o   = !from_lds ? sc.objects[isec.object] : *(const yhd_object*)(const YH_LDS yhd_object*)(tc.lds_scene + YH_OBJECT_F4 * isec.object);
  const yhd_material& mat = !from_lds ? sc.materials[o.material] : *(const yhd_material*)(const YH_LDS yhd_material*)(tc.lds_mats + YH_MATERIAL_F4 * o.material);
  hit_geom hg = eval_hit(sc, o, isec.slot, isec.u, isec.v);
  f3 position = hg.position;
  f3 nrm      = hg.normal;
  f3 normal;  // eval_shading_normal (pt.cpp:350-369)
  bool is_hair = o.kind == YH_KIND_LINES;
  if (is_hair) {
    normal = quad_orthonormalize(outgoing, nrm);
  } else {
    normal = (!mat.thin || dot(nrm, outgoing) >= 0) ? nrm : -nrm;
  }
  // Materials with lobes beyond diffuse / hair (dev_surface.h): the lobe mixture
  // decides opacity pass-through and whether the hit is a delta surface.
  if (COUNT) count_quad<COUNT>(is_hair ? tc.stats->hair : tc.stats->surf);
  const bool     general = GENERAL && !mat.plain;
  surface_brdf_t sb;
  float          tu = isec.u, tv = isec.v, etex_x = 1.0f;  // texcoord; emission texture, linear (pt.cpp:421)
  f3             ctex = mk3(1.0f), etex = mk3(1.0f);      // colour and emission texture values
  if (general) {
    if (mat.color_tex >= 0 || mat.emission_tex >= 0 || mat.scattering_tex >= 0) {
      eval_texcoord(sc, o, isec, tu, tv);
      ctex = eval_texture(sc, mat.color_tex, false, tu, tv);
      if (mat.emission_tex >= 0) {
        etex   = eval_texture(sc, mat.emission_tex, false, tu, tv);
        etex_x = eval_texture(sc, mat.emission_tex, true, tu, tv).x;
      }
    }
    sb = surface_brdf(mat, normal, outgoing, ctex, etex_x);
    if (sb.opacity < 1 && rand1f(rng) >= sb.opacity) {  // pt.cpp:1429-1433: pass through, same bounce
      ps.ray = mkray(position + ps.ray.d * 1e-2f, ps.ray.d);
      if (COUNT) tc.stats->c_rest += clock64() - k0;
      return true;
    }
  }
  ps.hit      = true;
  ps.radiance = ps.radiance + ps.weight * (ld3(mat.emission) * etex);
  if (general && is_delta(sb)) {  // pt.cpp:1452-1456 (also for a hair shape whose mixture is a delta)
    f3    incoming = surface_sample_delta(sb, normal, outgoing, rand1f(rng));
    f3    brdfcos;
    float pdf;
    surface_eval_pdf_delta(sb, normal, outgoing, incoming, brdfcos, pdf);
    ps.weight = ps.weight * (brdfcos / pdf);
    medium_crossing(sc, ps, mat, normal, outgoing, incoming, ctex, etex_x, tu, tv);
    ps.ray    = mkray(position, incoming);
    if (COUNT) tc.stats->c_rest += clock64() - k0;
    return path_continue(ps, rng, bounces);
  }
  hair_hit hh;
  hair_out ho;
  if (is_hair) {
    hh = hair_setup<!YH_LANE>(isec.v, normal, nrm);  // tangent = eval_normal (pt.cpp:487)
    ho = hair_prepare<!YH_LANE>(mat, hh, outgoing);  // shared by sample / eval / pdf
  }

  if (COUNT) k1 = clock64(), tc.stats->c_geom += k1 - k0;
  f3 incoming;
  if (rand1f(rng) < 0.5f) {
    float rnx = rand1f(rng), rny = rand1f(rng);
    float rnl = rand1f(rng);
    if (is_hair) {
      incoming = hair_sample(mat, hh, ho, rnx, rny);
    } else if (general) {
      incoming = surface_sample(sb, normal, outgoing, rnl, rnx, rny);
    } else {  // sample_brdfcos (pt.cpp:1139-1174): only the diffuse lobe
      incoming = mk3(0.0f);
      if (mat.diffuse_pdf != 0 && rnl < 0.0f + mat.diffuse_pdf) {
        if (!(dot(normal, outgoing) <= 0)) incoming = sample_hemisphere_cos(normal, rnx, rny);
      }
    }
  } else {
    float ruvx = rand1f(rng), ruvy = rand1f(rng);
    float rel = rand1f(rng);
    float rl  = rand1f(rng);
    incoming  = sample_lights<COUNT, GENERAL>(tc, position, rl, rel, ruvx, ruvy);
  }
  if (COUNT) k2 = clock64(), tc.stats->c_sample += k2 - k1;
  f3    brdfcos;
  float brdf_pdf;
  if (is_hair) {
#if YH_LANE
    hair_eval_pdf_lane(mat, hh, ho, incoming, brdfcos, brdf_pdf);
#else
    hair_eval_pdf_quad(mat, hh, ho, incoming, brdfcos, brdf_pdf);
#endif
  } else if (general) {
    surface_eval_pdf(sb, normal, outgoing, incoming, brdfcos, brdf_pdf);
  } else {  // eval_brdfcos / sample_brdfcos_pdf, diffuse lobe (math.h:4427,4572)
    brdfcos  = mk3(0.0f);
    brdf_pdf = 0.0f;
    f3 diffuse = ld3(mat.color);
    bool below = dot(normal, incoming) <= 0 || dot(normal, outgoing) <= 0;
    if (!is_zero(diffuse)) {
      f3 lobe = below ? mk3(0.0f) : mk3(1.0f) / pif * dot(normal, incoming);
      brdfcos = brdfcos + diffuse * lobe;
    }
    if (mat.diffuse_pdf != 0) {
      float lobe = 0.0f;
      if (!below) {
        float cosw = dot(normal, incoming);
        lobe       = (cosw <= 0) ? 0 : cosw / pif;
      }
      brdf_pdf += mat.diffuse_pdf * lobe;
    }
  }
  if (COUNT) k3 = clock64(), tc.stats->c_eval += k3 - k2;
#if YH_LANE
  float light_pdf = lane_lights_pdf<GENERAL>(tc, position, incoming);
#else
  float light_pdf = sample_lights_pdf<COUNT, STRIDE, GENERAL>(tc, position, incoming);
#endif
  ps.weight = ps.weight * (brdfcos / (0.5f * brdf_pdf + 0.5f * light_pdf));
  if (general) medium_crossing(sc, ps, mat, normal, outgoing, incoming, ctex, etex_x, tu, tv);
  ps.ray    = mkray(position, incoming);
  if (COUNT) tc.stats->c_rest += clock64() - k3;
  return path_continue(ps, rng, bounces);
}

// The reference's other shaders (get_trace_shader_func, pt.cpp:1660-1672), one loop iteration each
// like path_step: trace_naive (pt.cpp:1514-1581: brdf sampling only, no MIS, no volumes),
// trace_eyelight (pt.cpp:1584-1641: light at the eye, only delta chains continue, at least 4
// bounces) and trace_normal (pt.cpp:1644-1658: one hit, alpha 1 for hits and misses alike).
// Preview / debugging shaders: they always take the general lobe mixture (dev_surface.h).
template <bool COUNT, int STRIDE, int SHADER>
YH_DEV bool shade_step(const trace_ctx& tc, path_t& ps, const hit_t& isec, rng_t& rng, int bounces) {
  const yhd_scene& sc = *tc.sc;
  if (isec.object < 0) {
    f3 env = eval_environment<COUNT>(tc, ps.ray.d);
    if (SHADER == YH_SHADER_NORMAL) ps.radiance = env, ps.hit = true;
    else ps.radiance = ps.radiance + ps.weight * env;
    return false;
  }
  f3 outgoing = -ps.ray.d;
  const yhd_object&   o   = sc.objects[isec.object];
  const yhd_material& mat = sc.materials[o.material];
  hit_geom hg = eval_hit(sc, o, isec.slot, isec.u, isec.v);
  f3   position = hg.position, nrm = hg.normal;
  bool is_hair  = o.kind == YH_KIND_LINES;
  f3   normal   = is_hair ? quad_orthonormalize(outgoing, nrm) : ((!mat.thin || dot(nrm, outgoing) >= 0) ? nrm : -nrm);
  if (SHADER == YH_SHADER_NORMAL) {
    ps.radiance = normal * 0.5f + mk3(0.5f), ps.hit = true;
    return false;
  }
  if (COUNT) count_quad<COUNT>(is_hair ? tc.stats->hair : tc.stats->surf);
  float tu = isec.u, tv = isec.v, etex_x = 1.0f;
  f3    ctex = mk3(1.0f), etex = mk3(1.0f);
  if (mat.color_tex >= 0 || mat.emission_tex >= 0) {
    eval_texcoord(sc, o, isec, tu, tv);
    ctex = eval_texture(sc, mat.color_tex, false, tu, tv);
    if (mat.emission_tex >= 0) {
      etex   = eval_texture(sc, mat.emission_tex, false, tu, tv);
      etex_x = eval_texture(sc, mat.emission_tex, true, tu, tv).x;
    }
  }
  surface_brdf_t sb = surface_brdf(mat, normal, outgoing, ctex, etex_x);
  if (sb.opacity < 1 && rand1f(rng) >= sb.opacity) {  // pass through, same bounce
    ps.ray = mkray(position + ps.ray.d * 1e-2f, ps.ray.d);
    return true;
  }
  ps.hit      = true;
  ps.radiance = ps.radiance + ps.weight * (ld3(mat.emission) * etex);
  hair_hit hh;
  hair_out ho;
  if (is_hair) {
    hh = hair_setup<true>(isec.v, normal, nrm);
    ho = hair_prepare<true>(mat, hh, outgoing);
  }
  f3    incoming, brdfcos;
  float pdf;
  if (SHADER == YH_SHADER_EYELIGHT) {
    if (is_hair) hair_eval_pdf_quad(mat, hh, ho, outgoing, brdfcos, pdf);  // eval_brdfcos: hair first (pt.cpp:1071)
    else surface_eval_pdf(sb, normal, outgoing, outgoing, brdfcos, pdf);
    ps.radiance = ps.radiance + ps.weight * pif * brdfcos;
    if (!is_delta(sb)) return false;
    incoming = surface_sample_delta(sb, normal, outgoing, rand1f(rng));
    surface_eval_pdf_delta(sb, normal, outgoing, incoming, brdfcos, pdf);
    ps.weight = ps.weight * (brdfcos / pdf);
    if (is_zero(ps.weight) || !finite3(ps.weight)) return false;
    ps.ray = mkray(position, incoming);
    ps.bounce++;
    return ps.bounce < (bounces > 4 ? bounces : 4);
  }
  if (!is_delta(sb)) {
    // sample_brdfcos(brdf, normal, outgoing, rand1f(rng), rand2f(rng)): g++ draws rn first
    float rnx = rand1f(rng), rny = rand1f(rng);
    float rnl = rand1f(rng);
    if (is_hair) {
      incoming = hair_sample(mat, hh, ho, rnx, rny);
      hair_eval_pdf_quad(mat, hh, ho, incoming, brdfcos, pdf);
    } else {
      incoming = surface_sample(sb, normal, outgoing, rnl, rnx, rny);
      surface_eval_pdf(sb, normal, outgoing, incoming, brdfcos, pdf);
    }
  } else {
    incoming = surface_sample_delta(sb, normal, outgoing, rand1f(rng));
    surface_eval_pdf_delta(sb, normal, outgoing, incoming, brdfcos, pdf);
  }
  ps.weight = ps.weight * (brdfcos / pdf);
  ps.ray    = mkray(position, incoming);
  return path_continue(ps, rng, bounces);
}

// Start of trace_sample (pt.cpp:1676-1682): the four draws and the camera ray.
YH_DEV void path_begin(const yhd_camera& cam, path_t& ps, rng_t& rng, int i, int j, int w, int h) {
  float lu = rand1f(rng), lv = rand1f(rng);
  float pu = rand1f(rng), pv = rand1f(rng);
  ps.ray      = sample_camera(cam, i, j, w, h, pu, pv, lu, lv);
  ps.radiance = mk3(0.0f), ps.weight = mk3(1.0f);
  ps.bounce   = 0;
  ps.hit      = false;
  ps.in_medium = false;
}
// End of trace_sample (pt.cpp:1683-1686): sanitize, clamp, accumulate.
YH_DEV void path_end(const path_t& ps, float clamp, yhd_float4& acc) {
  f3 rgb = ps.radiance;
  if (!finite3(rgb)) rgb = mk3(0.0f);
  if (hmax(rgb) > clamp) rgb = rgb * (clamp / hmax(rgb));
  acc.x += rgb.x, acc.y += rgb.y, acc.z += rgb.z, acc.w += ps.hit ? 1.0f : 0.0f;
}

}  // namespace yhd
#endif
