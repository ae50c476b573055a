// ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" shim over the REAL reference (dsforza96/yocto-hair),
// compiled in this container from the sources where they lie under
// /root/reference (see oracle/Makefile; output only into oracle/_ref/).
// It exists so that (1) the CPU restatement in oracle/yh_oracle.cpp can be
// pinned bit-for-bit against the reference itself and (2) golden vectors can
// be generated (oracle/make_golden.py). Nothing in the product links this.
//
// It contains no reference source: it only calls the reference's public
// functions (yocto_extension.h:115-130, yocto_pathtrace.h:97-230,452-455,
// yocto_sceneio.h load_scene/get_camera, yocto_math.h inline helpers).
#include <yocto/yocto_math.h>
#include <yocto/yocto_sceneio.h>
#include <yocto/yocto_shape.h>
#include <yocto_extension/yocto_extension.h>
#include <yocto_pathtrace/yocto_pathtrace.h>

#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace ym  = yocto::math;
namespace ye  = yocto::extension;
namespace yp  = yocto::pathtrace;
namespace ysc = yocto::sceneio;

static ym::vec3f v3(const float* p) { return {p[0], p[1], p[2]}; }
static ym::ray3f mkray(const float* r) {
  return ym::ray3f{v3(r), v3(r + 3), r[6], r[7]};
}
static ye::hair_brdf mkbrdf(const float* p) {
  static_assert(sizeof(ye::hair_brdf) == 120, "hair_brdf layout");
  ye::hair_brdf b;
  std::memcpy((void*)&b, p, 120);
  return b;
}

struct ref_scene {
  std::unique_ptr<yp::scene> scene;
  yp::camera*                camera = nullptr;
};

extern "C" {

// math.h:1405-1442
void ref_rng_stream(uint64_t seed, uint64_t seq, int n, uint64_t* state_inc,
    float* out) {
  auto rng     = ym::make_rng(seed, seq);
  state_inc[0] = rng.state;
  state_inc[1] = rng.inc;
  for (int i = 0; i < n; i++) out[i] = ym::rand1f(rng);
}
// pt.cpp:1942-1945
void ref_pixel_seqs(int n, int* out) {
  auto rng = ym::make_rng(1301081);
  for (int i = 0; i < n; i++) out[i] = ym::rand1i(rng, 1 << 31) / 2 + 1;
}

// The public lobe functions of yocto_math.h:1513-1620, one kind per call
// (numbering = YH_LOBE_* in include/yhair.h). params 8n: ior, roughness,
// eta[3], etak[3]; rn 3n: rnl, rn.x, rn.y; out 7n: f*|cos| [3], pdf, sampled [3].
void ref_surface_lobe(int kind, int n, const float* params, const float* normal,
    const float* outgoing, const float* incoming, const float* rn, float* out) {
  for (int i = 0; i < n; i++) {
    auto q   = params + 8 * i;
    auto ior = q[0], rough = q[1];
    auto eta = v3(q + 2), etak = v3(q + 5);
    auto nn = v3(normal + 3 * i), wo = v3(outgoing + 3 * i), wi = v3(incoming + 3 * i);
    auto rnl = rn[3 * i];
    auto r2  = ym::vec2f{rn[3 * i + 1], rn[3 * i + 2]};
    auto f = ym::zero3f, w = ym::zero3f;
    auto pdf = 0.0f;
    switch (kind) {
      case 0:
        f = ym::eval_diffuse_reflection(nn, wo, wi), pdf = ym::sample_diffuse_reflection_pdf(nn, wo, wi);
        w = ym::sample_diffuse_reflection(nn, wo, r2);
        break;
      case 1:
        f   = ym::eval_microfacet_reflection(ior, rough, nn, wo, wi);
        pdf = ym::sample_microfacet_reflection_pdf(ior, rough, nn, wo, wi);
        w   = ym::sample_microfacet_reflection(ior, rough, nn, wo, r2);
        break;
      case 2:
        f   = ym::eval_microfacet_reflection(eta, etak, rough, nn, wo, wi);
        pdf = ym::sample_microfacet_reflection_pdf(eta, etak, rough, nn, wo, wi);
        w   = ym::sample_microfacet_reflection(eta, etak, rough, nn, wo, r2);
        break;
      case 3:
        f   = ym::eval_microfacet_transmission(ior, rough, nn, wo, wi);
        pdf = ym::sample_microfacet_transmission_pdf(ior, rough, nn, wo, wi);
        w   = ym::sample_microfacet_transmission(ior, rough, nn, wo, r2);
        break;
      case 4:
        f   = ym::eval_microfacet_refraction(ior, rough, nn, wo, wi);
        pdf = ym::sample_microfacet_refraction_pdf(ior, rough, nn, wo, wi);
        w   = ym::sample_microfacet_refraction(ior, rough, nn, wo, rnl, r2);
        break;
      case 5:
        f = ym::eval_delta_reflection(ior, nn, wo, wi), pdf = ym::sample_delta_reflection_pdf(ior, nn, wo, wi);
        w = ym::sample_delta_reflection(ior, nn, wo);
        break;
      case 6:
        f   = ym::eval_delta_reflection(eta, etak, nn, wo, wi);
        pdf = ym::sample_delta_reflection_pdf(eta, etak, nn, wo, wi);
        w   = ym::sample_delta_reflection(eta, etak, nn, wo);
        break;
      case 7:
        f = ym::eval_delta_transmission(ior, nn, wo, wi), pdf = ym::sample_delta_transmission_pdf(ior, nn, wo, wi);
        w = ym::sample_delta_transmission(ior, nn, wo);
        break;
      case 8:
        f = ym::eval_delta_refraction(ior, nn, wo, wi), pdf = ym::sample_delta_refraction_pdf(ior, nn, wo, wi);
        w = ym::sample_delta_refraction(ior, nn, wo, rnl);
        break;
      default: break;
    }
    auto o = out + 7 * i;
    o[0] = f.x, o[1] = f.y, o[2] = f.z, o[3] = pdf, o[4] = w.x, o[5] = w.y, o[6] = w.z;
  }
}
// The arithmetic of the pbrt curve conversion (yocto_pbrt.h:1751-1797) through the
// reference's own public helpers: interpolate_bezier, interpolate_bezier_derivative,
// lerp, normalize (yocto_math.h:1251-1257, 245, 2036).
void ref_curves_to_lines(int n, const float* P, const float* width0, const float* width1,
    int base_vertex, float* positions, float* normals, float* radius, int* lines) {
  auto number_sub = 4;
  for (int c = 0; c < n; c++) {
    auto p0 = v3(P + 12 * c), p1 = v3(P + 12 * c + 3), p2 = v3(P + 12 * c + 6), p3 = v3(P + 12 * c + 9);
    std::vector<ym::vec3f> pos, nrm;
    std::vector<float>     rad;
    pos.push_back(p0);
    for (auto i = 1; i < number_sub; i++) pos.push_back(ym::interpolate_bezier(p0, p1, p2, p3, (float)i / number_sub));
    pos.push_back(p3);
    nrm.push_back(ym::normalize(p1 - p0));
    for (auto i = 1; i < number_sub; i++)
      nrm.push_back(ym::normalize(ym::interpolate_bezier_derivative(p0, p1, p2, p3, (float)i / number_sub)));
    nrm.push_back(ym::normalize(p3 - p2));
    rad.push_back(width0[c]);
    for (auto i = 1; i < number_sub; i++) rad.push_back(ym::lerp(width0[c], width1[c], (float)i / number_sub));
    rad.push_back(width1[c]);
    for (int i = 0; i < 5; i++) {
      positions[15 * c + 3 * i] = pos[i].x, positions[15 * c + 3 * i + 1] = pos[i].y, positions[15 * c + 3 * i + 2] = pos[i].z;
      normals[15 * c + 3 * i] = nrm[i].x, normals[15 * c + 3 * i + 1] = nrm[i].y, normals[15 * c + 3 * i + 2] = nrm[i].z;
      radius[5 * c + i] = rad[i];
    }
    for (int i = 0; i < 4; i++) lines[8 * c + 2 * i] = base_vertex + 5 * c + i, lines[8 * c + 2 * i + 1] = base_vertex + 5 * c + i + 1;
  }
}
// fresnel_dielectric / fresnel_conductor / reflectivity_to_eta (yocto_math.h:1490-1503)
void ref_fresnel(int n, const float* params, const float* normal, const float* outgoing, float* out) {
  for (int i = 0; i < n; i++) {
    auto q  = params + 8 * i;
    auto nn = v3(normal + 3 * i), wo = v3(outgoing + 3 * i);
    auto fd = ym::fresnel_dielectric(q[0], nn, wo);
    auto fc = ym::fresnel_conductor(v3(q + 2), v3(q + 5), nn, wo);
    auto e  = ym::reflectivity_to_eta(v3(q + 2));
    auto o  = out + 7 * i;
    o[0] = fd, o[1] = fc.x, o[2] = fc.y, o[3] = fc.z, o[4] = e.x, o[5] = e.y, o[6] = e.z;
  }
}

// mats: 12 floats per item = sigma_a[3] beta_m beta_n alpha eta color[3]
// eumelanin pheomelanin (yocto_extension.h:86-95 field order)
void ref_hair_brdf(int n, const float* mats, const float* v, const float* nrm,
    const float* tng, float* out) {
  static_assert(sizeof(ye::hair_material) == 48, "hair_material layout");
  for (int i = 0; i < n; i++) {
    ye::hair_material m;
    std::memcpy((void*)&m, mats + 12 * i, 48);
    auto b = ye::eval_hair_brdf(m, v[i], v3(nrm + 3 * i), v3(tng + 3 * i));
    std::memcpy(out + 30 * i, &b, 120);
  }
}
void ref_hair_eval(
    int n, const float* brdf, const float* wo, const float* wi, float* out) {
  for (int i = 0; i < n; i++) {
    auto f = ye::eval_hair_scattering(
        mkbrdf(brdf + 30 * i), v3(wo + 3 * i), v3(wi + 3 * i));
    out[3 * i] = f.x, out[3 * i + 1] = f.y, out[3 * i + 2] = f.z;
  }
}
void ref_hair_sample(
    int n, const float* brdf, const float* wo, const float* rn, float* out) {
  for (int i = 0; i < n; i++) {
    auto w = ye::sample_hair_scattering(mkbrdf(brdf + 30 * i), v3(wo + 3 * i),
        ym::vec2f{rn[2 * i], rn[2 * i + 1]});
    out[3 * i] = w.x, out[3 * i + 1] = w.y, out[3 * i + 2] = w.z;
  }
}
void ref_hair_pdf(
    int n, const float* brdf, const float* wo, const float* wi, float* out) {
  for (int i = 0; i < n; i++)
    out[i] = ye::sample_hair_scattering_pdf(
        mkbrdf(brdf + 30 * i), v3(wo + 3 * i), v3(wi + 3 * i));
}
// yocto_extension.cpp:555-693; 1 = "OK!", 0 = "TEST FAILED!"
int ref_selftest(int which) {
  try {
    if (which == 0) ye::white_furnace_test();
    if (which == 1) ye::white_furnace_sampled_test();
    if (which == 2) ye::sampling_weights_test();
    if (which == 3) ye::sampling_consistency_test();
  } catch (...) {
    return 0;
  }
  return 1;
}

// math.h:3426-3505,3544-3554
void ref_intersect_line(int n, const float* rays, const float* p0,
    const float* p1, const float* r0, const float* r1, int* hit, float* uv,
    float* dist) {
  for (int i = 0; i < n; i++) {
    auto u = ym::vec2f{0, 0};
    auto d = 0.0f;
    hit[i] = ym::intersect_line(mkray(rays + 8 * i), v3(p0 + 3 * i),
        v3(p1 + 3 * i), r0[i], r1[i], u, d);
    uv[2 * i] = u.x, uv[2 * i + 1] = u.y, dist[i] = d;
  }
}
void ref_intersect_triangle(int n, const float* rays, const float* p0,
    const float* p1, const float* p2, int* hit, float* uv, float* dist) {
  for (int i = 0; i < n; i++) {
    auto u = ym::vec2f{0, 0};
    auto d = 0.0f;
    hit[i] = ym::intersect_triangle(mkray(rays + 8 * i), v3(p0 + 3 * i),
        v3(p1 + 3 * i), v3(p2 + 3 * i), u, d);
    uv[2 * i] = u.x, uv[2 * i + 1] = u.y, dist[i] = d;
  }
}
void ref_intersect_bbox(int n, const float* rays, const float* bbox, int* hit) {
  for (int i = 0; i < n; i++) {
    auto r    = mkray(rays + 8 * i);
    auto dinv = ym::vec3f{1 / r.d.x, 1 / r.d.y, 1 / r.d.z};
    hit[i]    = ym::intersect_bbox(
        r, dinv, ym::bbox3f{v3(bbox + 6 * i), v3(bbox + 6 * i + 3)});
  }
}

// Scene: load_scene + the CLI's sceneio->pathtrace conversion
// (apps/yscenetrace/yscenetrace.cpp:49-197, 225-247) through the public
// add_* / set_* API, then init_bvh + init_lights.
void* ref_scene_open(const char* json, const char* camera_name, char* err,
    int errlen) {
  auto io    = std::make_unique<ysc::model>();
  auto error = std::string{};
  if (!ysc::load_scene(json, io.get(), error)) {
    std::snprintf(err, errlen, "%s", error.c_str());
    return nullptr;
  }
  auto iocam = ysc::get_camera(io.get(), camera_name ? camera_name : "");
  auto rs    = new ref_scene{};
  rs->scene  = std::make_unique<yp::scene>();
  auto sc    = rs->scene.get();
  for (auto c : io->cameras) {
    auto cam = yp::add_camera(sc);
    yp::set_frame(cam, c->frame);
    yp::set_lens(cam, c->lens, c->aspect, c->film);
    yp::set_focus(cam, c->aperture, c->focus);
    if (c == iocam) rs->camera = cam;
  }
  std::unordered_map<ysc::texture*, yp::texture*> tex{{nullptr, nullptr}};
  for (auto t : io->textures) {
    auto o = yp::add_texture(sc);
    if (!t->colorf.empty()) yp::set_texture(o, t->colorf);
    else if (!t->colorb.empty()) yp::set_texture(o, t->colorb);
    else if (!t->scalarf.empty()) yp::set_texture(o, t->scalarf);
    else if (!t->scalarb.empty()) yp::set_texture(o, t->scalarb);
    tex[t] = o;
  }
  std::unordered_map<ysc::material*, yp::material*> mat{{nullptr, nullptr}};
  for (auto m : io->materials) {
    auto o = yp::add_material(sc);
    yp::set_eumelanin(o, m->eumelanin), yp::set_pheomelanin(o, m->pheomelanin);
    yp::set_sigma_a(o, m->sigma_a), yp::set_beta_m(o, m->beta_m);
    yp::set_beta_n(o, m->beta_n), yp::set_alpha(o, m->alpha);
    yp::set_eta(o, m->eta);
    yp::set_emission(o, m->emission, tex.at(m->emission_tex));
    yp::set_color(o, m->color, tex.at(m->color_tex));
    yp::set_specular(o, m->specular, tex.at(m->specular_tex));
    yp::set_ior(o, m->ior);
    yp::set_metallic(o, m->metallic, tex.at(m->metallic_tex));
    yp::set_transmission(o, m->transmission, m->thin, m->trdepth,
        tex.at(m->transmission_tex));
    yp::set_roughness(o, m->roughness, tex.at(m->roughness_tex));
    yp::set_opacity(o, m->opacity, tex.at(m->opacity_tex));
    yp::set_thin(o, m->thin);
    yp::set_scattering(
        o, m->scattering, m->scanisotropy, tex.at(m->scattering_tex));
    yp::set_normalmap(o, tex.at(m->normal_tex));
    mat[m] = o;
  }
  std::unordered_map<ysc::shape*, yp::shape*> shp{{nullptr, nullptr}};
  for (auto s : io->shapes) {
    auto o = yp::add_shape(sc);
    yp::set_points(o, s->points), yp::set_lines(o, s->lines);
    yp::set_triangles(o, s->triangles);
    if (!s->quads.empty())
      yp::set_triangles(o, yocto::shape::quads_to_triangles(s->quads));
    yp::set_positions(o, s->positions), yp::set_normals(o, s->normals);
    yp::set_texcoords(o, s->texcoords), yp::set_radius(o, s->radius);
    shp[s] = o;
  }
  for (auto ob : io->objects) {
    auto o = yp::add_object(sc);
    yp::set_frame(o, ob->frame);
    if (ob->shape) yp::set_shape(o, shp.at(ob->shape));
    yp::set_material(o, mat.at(ob->material));
  }
  for (auto e : io->environments) {
    auto o = yp::add_environment(sc);
    yp::set_frame(o, e->frame);
    yp::set_emission(o, e->emission, tex.at(e->emission_tex));
  }
  auto params = yp::trace_params{};
  yp::init_bvh(sc, params);
  yp::init_lights(sc, params, {});
  return rs;
}
void ref_scene_close(void* h) { delete (ref_scene*)h; }

// pt.cpp:1039-1046
void ref_scene_intersect(void* h, int n, const float* rays, int* object,
    int* element, float* uv, float* dist) {
  auto rs = (ref_scene*)h;
  for (int i = 0; i < n; i++) {
    auto isec  = yp::intersect_scene_bvh(rs->scene.get(), mkray(rays + 8 * i));
    object[i]  = isec.hit ? isec.object : -1;
    element[i] = isec.hit ? isec.element : -1;
    uv[2 * i] = isec.uv.x, uv[2 * i + 1] = isec.uv.y;
    dist[i] = isec.distance;
  }
}

// init_state + samples x trace_samples (cli.cpp:250-258). rgba may be NULL to
// query the size only. Returns 0 on success.
int ref_scene_render(void* h, int resolution, int samples, uint64_t seed,
    int bounces, float clamp, int noparallel, int* width, int* height,
    float* rgba, uint64_t* rng_state_inc, int shader) {
  auto rs           = (ref_scene*)h;
  auto params       = yp::trace_params{};
  params.resolution = resolution;
  params.samples    = samples;
  params.seed       = seed;
  params.bounces    = bounces;
  params.clamp      = clamp;
  params.noparallel = noparallel != 0;
  params.shader     = (yp::shader_type)shader;  // naive, path, eyelight, normal
  auto state        = std::make_unique<yp::state>();
  yp::init_state(state.get(), rs->scene.get(), rs->camera, params);
  auto size = state->render.size();
  *width = size.x, *height = size.y;
  if (!rgba) return 0;
  for (int s = 0; s < samples; s++)
    yp::trace_samples(state.get(), rs->scene.get(), rs->camera, params);
  std::memcpy(rgba, state->render.data(), sizeof(float) * 4 * size.x * size.y);
  if (rng_state_inc) {
    auto i = 0;
    for (auto& p : state->pixels) {
      rng_state_inc[i++] = p.rng.state;
      rng_state_inc[i++] = p.rng.inc;
    }
  }
  return 0;
}

int ref_scene_num_lights(void* h) {
  return (int)((ref_scene*)h)->scene->lights.size();
}
}  // extern "C"
