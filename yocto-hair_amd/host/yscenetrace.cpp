// yscenetrace — the reference's offline path tracer command line
// (apps/yscenetrace/yscenetrace.cpp:199-278) on the MI355X hair path.
//
// Same flags and defaults (cli.cpp:208-219): --camera, --resolution,-r 720,
// --samples,-s 512, --shader,-t path, --bounces,-b 8, --clamp 100,
// --save-batch, --output-image,-o out.hdr, positional scene. Extensions:
// --seed, --exact-bsdf, --device, --spp-per-launch, --gpus N / --devices A,B,.. (tile-sharded over
// the GPUs of one node, one RCCL gather of the float4 framebuffer at the end). Same flow: load scene -> convert through
// the add_* / set_* API -> init_bvh -> init_lights -> init_state -> sample loop
// -> save_image. Errors print and exit(1) like print_fatal
// (yocto_commonio.h:258-261).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "yscene_cli.h"

int main(int argc, const char* argv[]) {
  auto params = ptr::trace_params{};
  auto save_batch = false;
  std::string camera_name, imfilename = "out.hdr", filename, shader = "path";
  int  spp_per_launch = 256, gpus = 1, first_device = 0;  // (every launch waits for its unluckiest pixel: C1 0.2375 ms per sample at 64 per launch, 0.228 at 256, 0.2226 in one launch of 1536)
  std::string device_list;

  auto usage = [&]() {
    printf("usage: yscenetrace [--camera NAME] [--resolution,-r N] [--samples,-s N] [--shader,-t naive|path|eyelight|normal]\n"
           "                   [--bounces,-b N] [--clamp F] [--save-batch] [--output-image,-o FILE]\n"
           "                   [--seed N] [--exact-bsdf] [--device N] [--gpus N] [--devices A,B,..] [--spp-per-launch N] scene\n"
           "Offline path tracing of hair scenes on MI355X. --gpus N: the image's 8x8 tiles are dealt round-robin to N\n"
           "GPUs of this node (devices --device .. --device + N - 1, or --devices), one RCCL gather at the end.\n");
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
    else if (a == "--exact-bsdf") params.hair_exact = true;
    else if (a == "--device") first_device = atoi(next().c_str());
    else if (a == "--gpus") gpus = std::max(1, atoi(next().c_str()));
    else if (a == "--devices") device_list = next();
    else if (a == "--spp-per-launch") spp_per_launch = std::max(1, atoi(next().c_str()));
    else if (!a.empty() && a[0] == '-') print_fatal("unknown option " + a);
    else filename = a;
  }
  if (filename.empty()) { usage(); print_fatal("missing scene"); }
  set_devices(first_device, gpus, device_list);
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
    printf("render image: %d samples on %d GPU(s), %.3fs, %.1f Msamples/s\n", params.samples, (int)yhair::detail::devices().size(), dt,
        (double)state->width * state->height * params.samples / dt / 1e6);
    save(imfilename);
    printf("save image: %s\n", imfilename.c_str());
  } catch (const std::exception& e) {
    print_fatal(e.what());
  }
  return 0;
}
