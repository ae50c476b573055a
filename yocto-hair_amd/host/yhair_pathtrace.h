// yhair_pathtrace.h — host-side C++ mirror of the reference's interface for the
// hair path, implemented on top of the C ABI (include/yhair.h).
//
// Same names, argument meaning and error behaviour as
//   yocto::pathtrace  (libs/yocto_pathtrace/yocto_pathtrace.h:97-230): add_* /
//     set_* scene construction, trace_params, init_bvh, init_lights,
//     init_state, trace_samples;
//   yocto::extension  (libs/yocto_extension/yocto_extension.h:84-130):
//     hair_material, hair_brdf, eval_hair_brdf, eval_hair_scattering,
//     sample_hair_scattering, sample_hair_scattering_pdf (and its README name
//     eval_hair_scattering_pdf), and the four self-tests, which print "OK!" or
//     throw std::runtime_error("TEST FAILED!") exactly like the reference.
// so that a caller of the reference's API (apps/yscenetrace/yscenetrace.cpp:
// 241-258) switches by changing the namespace. All arithmetic runs on the GPU:
// these functions only marshal arguments. Programmer errors throw
// std::runtime_error (as pt.cpp:1669 does); there is no CPU fallback.
#ifndef YHAIR_PATHTRACE_H_
#define YHAIR_PATHTRACE_H_
#include <array>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "yhair.h"

namespace yhair::math {
struct vec2f { float x = 0, y = 0; };
struct vec2i { int x = 0, y = 0; };
struct vec3f { float x = 0, y = 0, z = 0; };
struct vec3i { int x = 0, y = 0, z = 0; };
struct vec4f { float x = 0, y = 0, z = 0, w = 0; };
struct frame3f {
  vec3f x = {1, 0, 0}, y = {0, 1, 0}, z = {0, 0, 1}, o = {0, 0, 0};
};
}  // namespace yhair::math

namespace yhair::detail {
// The devices the path runs on: one context per entry (default: device 0). With more than one, the image's
// 8x8 tiles are dealt round-robin to the contexts (yh_set_shard), every context renders all samples of its own
// tiles from its own host thread, and the framebuffer is gathered over RCCL (yh_gather_framebuffer). Pixel
// results do not depend on the number of GPUs. Set before the first call that needs a context.
inline std::vector<int>& devices() {
  static std::vector<int> devs = {0};
  return devs;
}
inline int& device() { return devices()[0]; }
inline std::vector<yh_context*>& contexts() {
  static std::vector<yh_context*> ctxs;
  return ctxs;
}
inline std::vector<yh_context*>& require_contexts() {
  auto& ctxs = contexts();
  if (ctxs.empty()) {
    for (int dev : devices()) {
      auto ctx = yh_create(dev);
      if (!ctx) throw std::runtime_error(std::string("yhair: ") + yh_last_error(nullptr));
      ctxs.push_back(ctx);
    }
  }
  return ctxs;
}
inline yh_context* require_context() { return require_contexts()[0]; }
inline yh_context* context() { return contexts().empty() ? nullptr : contexts()[0]; }
inline void check(int rc, yh_context* ctx = nullptr) {
  if (rc != YH_OK) throw std::runtime_error(std::string("yhair: ") + yh_last_error(ctx ? ctx : context()));
}
// fn(context, index) on every context, each from its own host thread (calls of the C ABI block)
template <typename F>
inline void for_each_context(F&& fn) {
  auto& ctxs = require_contexts();
  if (ctxs.size() == 1) {
    check(fn(ctxs[0], 0), ctxs[0]);
    return;
  }
  std::vector<int>         rcs(ctxs.size(), YH_OK);
  std::vector<std::thread> pool;
  for (size_t i = 0; i < ctxs.size(); i++) pool.emplace_back([&, i]() { rcs[i] = fn(ctxs[i], (int)i); });
  for (auto& t : pool) t.join();
  for (size_t i = 0; i < ctxs.size(); i++) check(rcs[i], ctxs[i]);
}
// the state whose pixels the contexts hold (see init_state / trace_samples)
inline const void*& bound_state() {
  static const void* st = nullptr;
  return st;
}
}  // namespace yhair::detail

// -----------------------------------------------------------------------------
namespace yhair::extension {
using math::frame3f;
using math::vec2f;
using math::vec3f;

inline const int p_max = 3;
struct hair_material {  // ext.h:86-95
  vec3f sigma_a     = {0, 0, 0};
  float beta_m      = 0.3f;
  float beta_n      = 0.3f;
  float alpha       = 2;
  float eta         = 1.55f;
  vec3f color       = {0, 0, 0};
  float eumelanin   = 0;
  float pheomelanin = 0;
};
struct hair_brdf {  // ext.h:97-113 (30 floats, same field order)
  vec3f                        sigma_a = {0, 0, 0};
  float                        alpha   = 2;
  float                        eta     = 1.55f;
  float                        h       = 0;
  std::array<float, p_max + 1> v       = {};
  float                        s       = 0;
  vec3f                        sin_2k_alpha, cos_2k_alpha;
  float                        gamma_o = 0;
  frame3f                      world_to_brdf;
};
static_assert(sizeof(hair_brdf) == sizeof(float) * YH_HAIR_BRDF_FLOATS, "hair_brdf layout");

inline yh_material to_material(const hair_material& m) {
  yh_material o{};
  o.opacity = 1, o.ior = 1.5f, o.thin = 1, o.trdepth = 0.01f;
  o.sigma_a[0] = m.sigma_a.x, o.sigma_a[1] = m.sigma_a.y, o.sigma_a[2] = m.sigma_a.z;
  o.beta_m = m.beta_m, o.beta_n = m.beta_n, o.alpha = m.alpha, o.eta = m.eta;
  o.color[0] = m.color.x, o.color[1] = m.color.y, o.color[2] = m.color.z;
  o.eumelanin = m.eumelanin, o.pheomelanin = m.pheomelanin;
  return o;
}
inline hair_brdf eval_hair_brdf(const hair_material& material, float v, const vec3f& normal, const vec3f& tangent) {
  auto      m = to_material(material);
  hair_brdf b;
  detail::check(yh_hair_brdf_batch(detail::require_context(), 1, &m, &v, &normal.x, &tangent.x, (float*)&b));
  return b;
}
inline vec3f eval_hair_scattering(const hair_brdf& brdf, const vec3f& outgoing, const vec3f& incoming) {
  vec3f f;
  detail::check(yh_hair_eval_batch(detail::require_context(), 1, (const float*)&brdf, &outgoing.x, &incoming.x, &f.x));
  return f;
}
inline vec3f sample_hair_scattering(const hair_brdf& brdf, const vec3f& outgoing, const vec2f& rn) {
  vec3f w;
  detail::check(yh_hair_sample_batch(detail::require_context(), 1, (const float*)&brdf, &outgoing.x, &rn.x, &w.x));
  return w;
}
inline float sample_hair_scattering_pdf(const hair_brdf& brdf, const vec3f& outgoing, const vec3f& incoming) {
  float pdf = 0;
  detail::check(yh_hair_pdf_batch(detail::require_context(), 1, (const float*)&brdf, &outgoing.x, &incoming.x, &pdf));
  return pdf;
}
// README.md:20 names it eval_hair_scattering_pdf
inline float eval_hair_scattering_pdf(const hair_brdf& brdf, const vec3f& outgoing, const vec3f& incoming) {
  return sample_hair_scattering_pdf(brdf, outgoing, incoming);
}
inline void run_selftest(int which) {
  auto rc = yh_selftest(detail::require_context(), which, nullptr);
  if (rc == YH_E_SELFTEST) throw std::runtime_error("TEST FAILED!");  // ext.cpp:582
  detail::check(rc);
  printf("OK!\n");
  fflush(stdout);
}
inline void white_furnace_test() { run_selftest(0); }
inline void white_furnace_sampled_test() { run_selftest(1); }
inline void sampling_weights_test() { run_selftest(2); }
inline void sampling_consistency_test() { run_selftest(3); }
}  // namespace yhair::extension

// -----------------------------------------------------------------------------
namespace yhair::pathtrace {
using math::frame3f;
using math::vec2f;
using math::vec2i;
using math::vec3f;
using math::vec3i;
using math::vec4f;

struct vec3b { unsigned char x = 0, y = 0, z = 0; };
struct texture {  // pt.h:282-287: the colour variants (colorf / colorb); scalar textures are not represented
  int                width = 0, height = 0;
  std::vector<vec3f> colorf;
  std::vector<vec3b> colorb;
};
struct camera {  // pt.h:272-278
  frame3f frame;
  float   lens     = 0.050f;
  vec2f   film     = {0.036f, 0.024f};
  float   focus    = 10000;
  float   aperture = 0;
};
struct material {  // pt.h:293-329 (of the textures: emission, colour, scattering)
  vec3f emission = {0, 0, 0}, color = {0, 0, 0};
  texture *emission_tex = nullptr, *color_tex = nullptr, *scattering_tex = nullptr;
  float specular = 0, roughness = 0, metallic = 0, ior = 1.5f, transmission = 0, opacity = 1;
  bool  thin = false;
  vec3f scattering = {0, 0, 0};
  float scanisotropy = 0, trdepth = 0.01f;
  float eumelanin = 0, pheomelanin = 0;
  vec3f sigma_a = {0, 0, 0};
  float beta_m = 0.3f, beta_n = 0.3f, alpha = 2, eta = 1.55f;
};
struct shape {  // pt.h:335-366
  std::vector<vec2i> lines;
  std::vector<vec3i> triangles;
  std::vector<vec3f> positions, normals;
  std::vector<vec2f> texcoords;
  std::vector<float> radius;
};
struct object {
  frame3f   frame;
  shape*    shape_   = nullptr;
  material* material_ = nullptr;
};
struct environment {
  frame3f  frame;
  vec3f    emission     = {0, 0, 0};
  texture* emission_tex = nullptr;
};
struct scene {
  std::vector<std::unique_ptr<camera>>      cameras;
  std::vector<std::unique_ptr<object>>      objects;
  std::vector<std::unique_ptr<shape>>       shapes;
  std::vector<std::unique_ptr<material>>    materials;
  std::vector<std::unique_ptr<texture>>     textures;
  std::vector<std::unique_ptr<environment>> environments;
  // set by init_bvh / init_lights, consumed by init_state (which knows the camera)
  mutable bool          bvh_requested = false, lights_requested = false;
  mutable const camera* uploaded_for  = nullptr;
  mutable double        upload_seconds = 0;  // wall-clock of the last flatten + yh_upload_scene (init_bvh + init_lights of the reference), for the command line's --timing
};
struct state {  // pt.h:426-429; `render` is refreshed by trace_samples
  int                width = 0, height = 0, samples = 0;
  std::vector<vec4f> render;
  yh_trace_params    device_params{};  // what init_state asked for (see trace_samples: re-binding)
};
enum struct shader_type { naive, path, eyelight, normal };
const auto default_seed = 961748941ull;
struct trace_params {  // pt.h:188-197
  int         resolution = 720;
  shader_type shader     = shader_type::path;
  int         samples    = 512;
  int         bounces    = 8;
  float       clamp      = 100;
  uint64_t    seed       = default_seed;
  bool        noparallel = false;
  int         pratio     = 8;
  bool        hair_exact = false;  // extension: the hair BSDF's exact arithmetic (yh_trace_params::hair_exact)
};
const auto shader_names = std::vector<std::string>{"naive", "path", "eyelight", "normal"};
using progress_callback = std::function<void(const std::string& message, int current, int total)>;

// scene construction (pt.h:97-174)
inline camera*      add_camera(scene* s) { return s->cameras.emplace_back(new camera{}).get(); }
inline object*      add_object(scene* s) { return s->objects.emplace_back(new object{}).get(); }
inline texture*     add_texture(scene* s) { return s->textures.emplace_back(new texture{}).get(); }
inline material*    add_material(scene* s) { return s->materials.emplace_back(new material{}).get(); }
inline shape*       add_shape(scene* s) { return s->shapes.emplace_back(new shape{}).get(); }
inline environment* add_environment(scene* s) { return s->environments.emplace_back(new environment{}).get(); }
inline void set_frame(camera* c, const frame3f& f) { c->frame = f; }
inline void set_lens(camera* c, float lens, float aspect, float film) {  // pt.cpp:2088-2092
  c->lens = lens;
  c->film = aspect >= 1 ? vec2f{film, film / aspect} : vec2f{film * aspect, film};
}
inline void set_focus(camera* c, float aperture, float focus) { c->aperture = aperture, c->focus = focus; }
inline void set_frame(object* o, const frame3f& f) { o->frame = f; }
inline void set_material(object* o, material* m) { o->material_ = m; }
inline void set_shape(object* o, shape* s) { o->shape_ = s; }
inline void set_texture(texture* t, int width, int height, const std::vector<vec3f>& img) {
  t->width = width, t->height = height, t->colorf = img;
}
inline void set_eumelanin(material* m, float v) { m->eumelanin = v; }
inline void set_pheomelanin(material* m, float v) { m->pheomelanin = v; }
inline void set_sigma_a(material* m, vec3f v) { m->sigma_a = v; }
inline void set_beta_m(material* m, float v) { m->beta_m = v; }
inline void set_beta_n(material* m, float v) { m->beta_n = v; }
inline void set_alpha(material* m, float v) { m->alpha = v; }
inline void set_eta(material* m, float v) { m->eta = v; }
inline void set_texture(texture* t, int width, int height, const std::vector<vec3b>& img) {
  t->width = width, t->height = height, t->colorb = img, t->colorf.clear();
}
inline void set_emission(material* m, const vec3f& e, texture* tex = nullptr) { m->emission = e, m->emission_tex = tex; }
inline void set_color(material* m, const vec3f& c, texture* tex = nullptr) { m->color = c, m->color_tex = tex; }
inline void set_texcoords(shape* s, const std::vector<vec2f>& v) { s->texcoords = v; }
inline void set_specular(material* m, float v = 1) { m->specular = v; }
inline void set_ior(material* m, float v) { m->ior = v; }
inline void set_metallic(material* m, float v) { m->metallic = v; }
inline void set_transmission(material* m, float t, bool thin, float trdepth) {
  m->transmission = t, m->thin = thin, m->trdepth = trdepth;
}
inline void set_scattering(material* m, const vec3f& scattering, float scanisotropy, texture* tex = nullptr) {
  m->scattering = scattering, m->scanisotropy = scanisotropy, m->scattering_tex = tex;
}
inline void set_roughness(material* m, float v) { m->roughness = v; }
inline void set_opacity(material* m, float v) { m->opacity = v; }
inline void set_thin(material* m, bool thin) { m->thin = thin; }
inline void set_lines(shape* s, const std::vector<vec2i>& v) { s->lines = v; }
inline void set_triangles(shape* s, const std::vector<vec3i>& v) { s->triangles = v; }
inline void set_positions(shape* s, const std::vector<vec3f>& v) { s->positions = v; }
inline void set_normals(shape* s, const std::vector<vec3f>& v) { s->normals = v; }
inline void set_radius(shape* s, const std::vector<float>& v) { s->radius = v; }
// (the same taking a temporary: the reference's signatures copy; a caller that hands over a vector it no longer needs — a loader — moves it: 60 MB for a hair model)
inline void set_lines(shape* s, std::vector<vec2i>&& v) { s->lines = std::move(v); }
inline void set_triangles(shape* s, std::vector<vec3i>&& v) { s->triangles = std::move(v); }
inline void set_positions(shape* s, std::vector<vec3f>&& v) { s->positions = std::move(v); }
inline void set_normals(shape* s, std::vector<vec3f>&& v) { s->normals = std::move(v); }
inline void set_radius(shape* s, std::vector<float>&& v) { s->radius = std::move(v); }
inline void set_frame(environment* e, const frame3f& f) { e->frame = f; }
inline void set_emission(environment* e, const vec3f& em, texture* tex = nullptr) { e->emission = em, e->emission_tex = tex; }

// Flattens the scene graph into a yh_scene_desc and uploads it; the C ABI builds
// the BVHs (init_bvh, pt.cpp:755-818) and the lights (init_lights,
// pt.cpp:1695-1740) in that one call. The camera is part of the uploaded scene,
// so the upload happens in init_state, the first call that receives it.
inline void upload_scene(const scene* sc, const camera* cam) {
  std::vector<yh_shape>       shapes;
  std::vector<yh_material>    materials;
  std::vector<yh_object>      objects;
  std::vector<yh_environment> envs;
  std::vector<yh_texture>     textures;
  std::vector<int>            texture_slot(sc->textures.size(), 0);  // 1-based slot in `textures`, 0 = not a material texture
  auto material_texture = [&](const texture* t) -> int {
    if (!t) return 0;
    for (size_t i = 0; i < sc->textures.size(); i++) {
      if (sc->textures[i].get() != t) continue;
      if (!texture_slot[i]) {
        yh_texture o{};
        o.width = t->width, o.height = t->height, o.is_byte = t->colorf.empty();
        o.pixels = o.is_byte ? (const void*)t->colorb.data() : (const void*)t->colorf.data();
        textures.push_back(o);
        texture_slot[i] = (int)textures.size();
      }
      return texture_slot[i];
    }
    throw std::runtime_error("yhair: material references a texture that is not in the scene");
  };
  auto index_of = [](auto& vec, auto* p) {
    for (size_t i = 0; i < vec.size(); i++)
      if (vec[i].get() == p) return (int)i;
    throw std::runtime_error("yhair: object references a shape/material that is not in the scene");
  };
  for (auto& s : sc->shapes) {
    yh_shape o{};
    o.num_vertices  = (int)s->positions.size();
    o.positions     = (const float*)s->positions.data();
    o.normals       = s->normals.empty() ? nullptr : (const float*)s->normals.data();
    o.radius        = s->radius.empty() ? nullptr : s->radius.data();
    o.num_lines     = (int)s->lines.size();
    o.lines         = s->lines.empty() ? nullptr : (const int*)s->lines.data();
    o.num_triangles = s->lines.empty() ? (int)s->triangles.size() : 0;
    o.triangles     = o.num_triangles ? (const int*)s->triangles.data() : nullptr;
    o.texcoords     = s->texcoords.size() == s->positions.size() && !s->texcoords.empty() ? (const float*)s->texcoords.data() : nullptr;
    shapes.push_back(o);
  }
  for (auto& m : sc->materials) {
    yh_material o{};
    o.emission[0] = m->emission.x, o.emission[1] = m->emission.y, o.emission[2] = m->emission.z;
    o.color[0] = m->color.x, o.color[1] = m->color.y, o.color[2] = m->color.z;
    o.specular = m->specular, o.metallic = m->metallic, o.roughness = m->roughness;
    o.transmission = m->transmission, o.opacity = m->opacity, o.ior = m->ior, o.thin = m->thin;
    o.sigma_a[0] = m->sigma_a.x, o.sigma_a[1] = m->sigma_a.y, o.sigma_a[2] = m->sigma_a.z;
    o.beta_m = m->beta_m, o.beta_n = m->beta_n, o.alpha = m->alpha, o.eta = m->eta;
    o.eumelanin = m->eumelanin, o.pheomelanin = m->pheomelanin;
    o.scattering[0] = m->scattering.x, o.scattering[1] = m->scattering.y, o.scattering[2] = m->scattering.z;
    o.scanisotropy = m->scanisotropy, o.trdepth = m->trdepth;
    o.emission_tex = material_texture(m->emission_tex), o.color_tex = material_texture(m->color_tex);
    o.scattering_tex = material_texture(m->scattering_tex);
    materials.push_back(o);
  }
  for (auto& ob : sc->objects) {
    yh_object o{};
    static_assert(sizeof(frame3f) == 48, "frame3f layout");
    memcpy(o.frame, &ob->frame, 48);
    o.shape = index_of(sc->shapes, ob->shape_), o.material = index_of(sc->materials, ob->material_);
    objects.push_back(o);
  }
  for (auto& e : sc->environments) {
    yh_environment o{};
    memcpy(o.frame, &e->frame, 48);
    o.emission[0] = e->emission.x, o.emission[1] = e->emission.y, o.emission[2] = e->emission.z;
    if (e->emission_tex) {
      o.tex_width = e->emission_tex->width, o.tex_height = e->emission_tex->height;
      o.texels = (const float*)e->emission_tex->colorf.data();
    }
    envs.push_back(o);
  }
  yh_scene_desc d{};
  d.num_shapes = (int)shapes.size(), d.shapes = shapes.data();
  d.num_materials = (int)materials.size(), d.materials = materials.data();
  d.num_objects = (int)objects.size(), d.objects = objects.data();
  d.num_environments = (int)envs.size(), d.environments = envs.data();
  d.num_textures = (int)textures.size(), d.textures = textures.data();
  if (!cam) throw std::runtime_error("yhair: no camera");
  memcpy(d.camera.frame, &cam->frame, 48);
  d.camera.lens = cam->lens, d.camera.film[0] = cam->film.x, d.camera.film[1] = cam->film.y;
  d.camera.focus = cam->focus, d.camera.aperture = cam->aperture;
  detail::for_each_context([&](yh_context* ctx, int) { return yh_upload_scene(ctx, &d); });  // the scene is replicated
  sc->uploaded_for = cam;
}
// init_bvh / init_lights: same signatures as pt.h:207-217. They mark the scene;
// the build itself runs inside the upload (see upload_scene).
inline void init_bvh(scene* sc, const trace_params&, progress_callback progress_cb = {}) {
  sc->bvh_requested = true, sc->uploaded_for = nullptr;
  if (progress_cb) progress_cb("build bvh", 1, 1);
}
inline void init_lights(scene* sc, const trace_params&, progress_callback progress_cb = {}) {
  sc->lights_requested = true, sc->uploaded_for = nullptr;
  if (progress_cb) progress_cb("build light", 1, 1);
}
// init_state (pt.cpp:1931-1946)
inline void init_state(state* st, const scene* sc, const camera* cam, const trace_params& params) {
  if ((int)params.shader < 0 || (int)params.shader > (int)shader_type::normal)
    throw std::runtime_error("sampler unknown");  // pt.cpp:1669
  if (!sc->bvh_requested || !sc->lights_requested)
    throw std::runtime_error("yhair: init_state before init_bvh / init_lights");
  if (sc->uploaded_for != cam) {
    const auto t0 = std::chrono::steady_clock::now();
    upload_scene(sc, cam);
    sc->upload_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  st->device_params = yh_trace_params{params.resolution, params.bounces, params.clamp, params.seed, (int)params.shader, params.hair_exact ? 1 : 0};
  detail::for_each_context([&](yh_context* ctx, int i) {
    int rc = yh_set_shard(ctx, i, (int)detail::contexts().size());
    return rc ? rc : yh_init_state(ctx, &st->device_params);
  });
  detail::check(yh_image_size(detail::require_context(), &st->width, &st->height));
  st->samples = 0;
  st->render.assign((size_t)st->width * st->height, vec4f{});
  detail::bound_state() = st;
}
// The contexts hold the pixels of ONE state. The reference's interactive caller initialises its render state,
// then initialises and traces a low-resolution preview state, then traces the render state
// (apps/ysceneitraces/ysceneitraces.cpp:255-300): a state that was displaced by another init_state before it
// accumulated any sample is simply initialised again on the device; one that had samples cannot be.
inline void bind_state(state* st) {
  if (detail::bound_state() == st) return;
  if (st->samples != 0) throw std::runtime_error("yhair: this state's pixels were displaced by another init_state");
  detail::for_each_context([&](yh_context* ctx, int) { return yh_init_state(ctx, &st->device_params); });
  detail::bound_state() = st;
}
inline void download(state* st) {
  auto& ctxs = detail::require_contexts();
  if (ctxs.size() == 1) detail::check(yh_download(ctxs[0], (float*)st->render.data()));
  else detail::check(yh_gather_framebuffer(ctxs.data(), (int)ctxs.size(), (float*)st->render.data()));
}
// trace_samples (pt.cpp:1992-2007): `nsamples` calls of the reference's
// function in one launch per GPU; state->render is refreshed when `download` is set.
inline void trace_samples(state* st, const scene*, const camera*, const trace_params&, int nsamples = 1,
    bool download_image = true) {
  bind_state(st);
  detail::for_each_context([&](yh_context* ctx, int) { return yh_trace_samples(ctx, nsamples); });
  st->samples += nsamples;
  if (download_image) download(st);
}
// the stop-flag overload (pt.cpp:2009-2026): one sample per call like the reference's; the flag is looked at
// before the launch and again before the image is copied back, so a set flag costs at most one launch.
inline void trace_samples(state* st, const scene* sc, const camera* cam, const trace_params& params,
    std::atomic<bool>* stop) {
  if (stop && *stop) return;
  trace_samples(st, sc, cam, params, 1, false);
  if (stop && *stop) return;
  download(st);
}
}  // namespace yhair::pathtrace
#endif
