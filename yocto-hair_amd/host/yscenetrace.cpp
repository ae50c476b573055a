// yscenetrace — the reference's offline path tracer command line
// (apps/yscenetrace/yscenetrace.cpp:199-278) on the MI355X hair path.
//
// Same flags and defaults (cli.cpp:208-219): --camera, --resolution,-r 720,
// --samples,-s 512, --shader,-t path, --bounces,-b 8, --clamp 100,
// --save-batch, --output-image,-o out.hdr, positional scene. Extensions:
// --seed, --device, --spp-per-launch. Same flow: load scene -> convert through
// the add_* / set_* API -> init_bvh -> init_lights -> init_state -> sample loop
// -> save_image. Errors print and exit(1) like print_fatal
// (yocto_commonio.h:258-261).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "yhair_pathtrace.h"

namespace ptr = yhair::pathtrace;
using namespace yhair::math;

[[noreturn]] static void print_fatal(const std::string& msg) {
  printf("%s\n", msg.c_str());
  exit(1);
}

// sio::model -> ptr::scene through the public scene-construction API, as
// init_scene does in the reference CLI (cli.cpp:49-197).
static ptr::camera* init_scene(ptr::scene* scene, const yh_scene_desc* d) {
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
    auto  v3 = [](const float* p, int n) {
      std::vector<vec3f> v(n);
      if (n) memcpy(v.data(), p, sizeof(float) * 3 * n);
      return v;
    };
    ptr::set_positions(o, v3(s.positions, s.num_vertices));
    if (s.normals) ptr::set_normals(o, v3(s.normals, s.num_vertices));
    if (s.radius) ptr::set_radius(o, std::vector<float>(s.radius, s.radius + s.num_vertices));
    if (s.texcoords) {
      std::vector<yhair::pathtrace::vec2f> tc((size_t)s.num_vertices);
      memcpy((void*)tc.data(), s.texcoords, sizeof(float) * 2 * (size_t)s.num_vertices);
      ptr::set_texcoords(o, tc);
    }
    if (s.num_lines) {
      std::vector<vec2i> l(s.num_lines);
      memcpy(l.data(), s.lines, sizeof(int) * 2 * s.num_lines);
      ptr::set_lines(o, l);
    }
    if (s.num_triangles) {
      std::vector<vec3i> t(s.num_triangles);
      memcpy(t.data(), s.triangles, sizeof(int) * 3 * s.num_triangles);
      ptr::set_triangles(o, t);
    }
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

int main(int argc, const char* argv[]) {
  auto params = ptr::trace_params{};
  auto save_batch = false;
  std::string camera_name, imfilename = "out.hdr", filename, shader = "path";
  int  spp_per_launch = 64;

  auto usage = [&]() {
    printf("usage: yscenetrace [--camera NAME] [--resolution,-r N] [--samples,-s N] [--shader,-t naive|path|eyelight|normal]\n"
           "                   [--bounces,-b N] [--clamp F] [--save-batch] [--output-image,-o FILE]\n"
           "                   [--seed N] [--device N] [--spp-per-launch N] scene\n"
           "Offline path tracing of hair scenes on MI355X\n");
  };
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> std::string {
      if (i + 1 >= argc) print_fatal("missing value for " + a);
      return argv[++i];
    };
    if (a == "--help" || a == "-h") { usage(); return 0; }
    else if (a == "--camera") camera_name = next();
    else if (a == "--resolution" || a == "-r") params.resolution = atoi(next().c_str());
    else if (a == "--samples" || a == "-s") params.samples = atoi(next().c_str());
    else if (a == "--shader" || a == "-t") shader = next();
    else if (a == "--bounces" || a == "-b") params.bounces = atoi(next().c_str());
    else if (a == "--clamp") params.clamp = (float)atof(next().c_str());
    else if (a == "--save-batch") save_batch = true;
    else if (a == "--output-image" || a == "-o") imfilename = next();
    else if (a == "--seed") params.seed = strtoull(next().c_str(), nullptr, 10);
    else if (a == "--device") yhair::detail::device() = atoi(next().c_str());
    else if (a == "--spp-per-launch") spp_per_launch = std::max(1, atoi(next().c_str()));
    else if (!a.empty() && a[0] == '-') print_fatal("unknown option " + a);
    else filename = a;
  }
  if (filename.empty()) { usage(); print_fatal("missing scene"); }
  bool known = false;
  for (size_t i = 0; i < ptr::shader_names.size(); i++)
    if (ptr::shader_names[i] == shader) params.shader = (ptr::shader_type)i, known = true;
  if (!known) print_fatal("unknown shader " + shader);

  try {
    char error[512];
    auto t0 = std::chrono::steady_clock::now();
    auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    auto ioscene = yh_scene_load(filename.c_str(), camera_name.c_str(), error, sizeof(error));
    if (!ioscene) print_fatal(error);
    printf("load scene: %.2fs\n", secs());

    auto scene  = std::make_unique<ptr::scene>();
    auto camera = init_scene(scene.get(), yh_scene_get(ioscene));
    yh_scene_free(ioscene);

    ptr::init_bvh(scene.get(), params);
    ptr::init_lights(scene.get(), params);
    auto state = std::make_unique<ptr::state>();
    ptr::init_state(state.get(), scene.get(), camera, params);
    printf("build bvh + lights + state (%dx%d): %.2fs\n", state->width, state->height, secs());

    auto save = [&](const std::string& name) {
      if (yh_save_image(name.c_str(), state->width, state->height, (const float*)state->render.data(), error,
              sizeof(error)) != YH_OK)
        print_fatal(error);
    };
    auto t1 = std::chrono::steady_clock::now();
    for (int sample = 0; sample < params.samples;) {
      int n = save_batch ? 1 : std::min(spp_per_launch, params.samples - sample);
      bool last = sample + n >= params.samples;
      ptr::trace_samples(state.get(), scene.get(), camera, params, n, save_batch || last);
      sample += n;
      if (save_batch) {  // cli.cpp:259-267: "<stem>-s<sample><ext>"
        auto dot_pos = imfilename.rfind('.');
        auto stem = imfilename.substr(0, dot_pos), ext = dot_pos == std::string::npos ? "" : imfilename.substr(dot_pos);
        save(stem + "-s" + std::to_string(sample - 1) + ext);
      }
    }
    if (params.samples == 0) ptr::trace_samples(state.get(), scene.get(), camera, params, 0, true);
    auto dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    printf("render image: %d samples, %.3fs, %.1f Msamples/s\n", params.samples, dt,
        (double)state->width * state->height * params.samples / dt / 1e6);
    save(imfilename);
    printf("save image: %s\n", imfilename.c_str());
  } catch (const std::exception& e) {
    print_fatal(e.what());
  }
  return 0;
}
