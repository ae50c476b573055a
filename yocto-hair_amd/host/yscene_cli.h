// yscene_cli.h — what the two command lines share: print_fatal (yocto_commonio.h:258-261) and init_scene, the
// conversion of a loaded scene into a ptr::scene through the public add_* / set_* API, as both of the
// reference's apps do (apps/yscenetrace/yscenetrace.cpp:49-197, apps/ysceneitraces/ysceneitraces.cpp:96-243).
#ifndef YSCENE_CLI_H_
#define YSCENE_CLI_H_
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "yhair_pathtrace.h"

namespace ptr = yhair::pathtrace;
using namespace yhair::math;

[[noreturn]] inline void print_fatal(const std::string& msg) {
  printf("%s\n", msg.c_str());
  exit(1);
}

// sio::model -> ptr::scene through the public scene-construction API, as
// init_scene does in the reference CLI (cli.cpp:49-197).
inline ptr::camera* init_scene(ptr::scene* scene, const yh_scene_desc* d) {
  auto frame_of = [](const float* f) {
    frame3f r;
    memcpy(&r, f, 48);
    return r;
  };
  auto camera = ptr::add_camera(scene);
  ptr::set_frame(camera, frame_of(d->camera.frame));
  camera->lens = d->camera.lens, camera->film = {d->camera.film[0], d->camera.film[1]};
  ptr::set_focus(camera, d->camera.aperture, d->camera.focus);
  std::vector<ptr::texture*> textures;  // material colour textures (yh_scene_desc::textures)
  for (int i = 0; i < d->num_textures; i++) {
    auto&  t = d->textures[i];
    auto   o = ptr::add_texture(scene);
    size_t n = (size_t)t.width * t.height;
    if (t.is_byte) {
      std::vector<yhair::pathtrace::vec3b> img(n);
      memcpy((void*)img.data(), t.pixels, 3 * n);
      ptr::set_texture(o, t.width, t.height, img);
    } else {
      std::vector<vec3f> img(n);
      memcpy((void*)img.data(), t.pixels, sizeof(float) * 3 * n);
      ptr::set_texture(o, t.width, t.height, img);
    }
    textures.push_back(o);
  }
  auto texture_of = [&](int id) { return id > 0 ? textures[(size_t)id - 1] : nullptr; };
  std::vector<ptr::material*> materials;
  for (int i = 0; i < d->num_materials; i++) {
    auto& m = d->materials[i];
    auto  o = ptr::add_material(scene);
    ptr::set_eumelanin(o, m.eumelanin), ptr::set_pheomelanin(o, m.pheomelanin);
    ptr::set_sigma_a(o, {m.sigma_a[0], m.sigma_a[1], m.sigma_a[2]});
    ptr::set_beta_m(o, m.beta_m), ptr::set_beta_n(o, m.beta_n), ptr::set_alpha(o, m.alpha), ptr::set_eta(o, m.eta);
    ptr::set_emission(o, vec3f{m.emission[0], m.emission[1], m.emission[2]}, texture_of(m.emission_tex));
    ptr::set_color(o, {m.color[0], m.color[1], m.color[2]}, texture_of(m.color_tex));
    ptr::set_specular(o, m.specular), ptr::set_ior(o, m.ior), ptr::set_metallic(o, m.metallic);
    ptr::set_transmission(o, m.transmission, m.thin != 0, m.trdepth);
    ptr::set_scattering(o, {m.scattering[0], m.scattering[1], m.scattering[2]}, m.scanisotropy, texture_of(m.scattering_tex));
    ptr::set_roughness(o, m.roughness), ptr::set_opacity(o, m.opacity), ptr::set_thin(o, m.thin != 0);
    materials.push_back(o);
  }
  std::vector<ptr::shape*> shapes;
  for (int i = 0; i < d->num_shapes; i++) {
    auto& s = d->shapes[i];
    auto  o = ptr::add_shape(scene);
    auto  v3 = [](const float* p, int n) {  // (one pass: no zero fill before the copy)
      static_assert(sizeof(vec3f) == 12, "vec3f is three floats");
      const vec3f* q = (const vec3f*)p;
      return std::vector<vec3f>(q, q + n);
    };
    ptr::set_positions(o, v3(s.positions, s.num_vertices));
    if (s.normals) ptr::set_normals(o, v3(s.normals, s.num_vertices));
    if (s.radius) ptr::set_radius(o, std::vector<float>(s.radius, s.radius + s.num_vertices));
    if (s.texcoords) {
      std::vector<yhair::pathtrace::vec2f> tc((size_t)s.num_vertices);
      memcpy((void*)tc.data(), s.texcoords, sizeof(float) * 2 * (size_t)s.num_vertices);
      ptr::set_texcoords(o, tc);
    }
    if (s.num_lines) ptr::set_lines(o, std::vector<vec2i>((const vec2i*)s.lines, (const vec2i*)s.lines + s.num_lines));
    if (s.num_triangles) ptr::set_triangles(o, std::vector<vec3i>((const vec3i*)s.triangles, (const vec3i*)s.triangles + s.num_triangles));
    shapes.push_back(o);
  }
  for (int i = 0; i < d->num_objects; i++) {
    auto o = ptr::add_object(scene);
    ptr::set_frame(o, frame_of(d->objects[i].frame));
    ptr::set_shape(o, shapes[d->objects[i].shape]);
    ptr::set_material(o, materials[d->objects[i].material]);
  }
  for (int i = 0; i < d->num_environments; i++) {
    auto& e = d->environments[i];
    auto  o = ptr::add_environment(scene);
    ptr::set_frame(o, frame_of(e.frame));
    ptr::texture* tex = nullptr;
    if (e.texels) {
      tex = ptr::add_texture(scene);
      std::vector<vec3f> img((size_t)e.tex_width * e.tex_height);
      memcpy(img.data(), e.texels, sizeof(float) * 3 * img.size());
      ptr::set_texture(tex, e.tex_width, e.tex_height, img);
    }
    ptr::set_emission(o, {e.emission[0], e.emission[1], e.emission[2]}, tex);
  }
  return camera;
}

// "--devices a,b,c" / "--gpus N": the devices the contexts are made on (yhair_pathtrace.h: detail::devices())
inline void set_devices(int first, int gpus, const std::string& list) {
  auto& devs = yhair::detail::devices();
  devs.clear();
  if (!list.empty()) {
    size_t at = 0;
    while (at <= list.size()) {
      size_t comma = list.find(',', at);
      if (comma == std::string::npos) comma = list.size();
      devs.push_back(atoi(list.substr(at, comma - at).c_str()));
      at = comma + 1;
    }
  } else {
    for (int i = 0; i < (gpus < 1 ? 1 : gpus); i++) devs.push_back(first + i);
  }
}
#endif
