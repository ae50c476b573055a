// yscenetrace — the reference's offline path tracer command line
// (apps/yscenetrace/yscenetrace.cpp:199-278) on the MI355X hair path.
//
// Same flags and defaults (cli.cpp:208-219): --camera, --resolution,-r 720,
// --samples,-s 512, --shader,-t path, --bounces,-b 8, --clamp 100,
// --save-batch, --output-image,-o out.hdr, positional scene. Extensions:
// --seed, --exact-bsdf, --device, --spp-per-launch, --timing (one "timing: {json}" line with the wall-clock of every phase: what
// bench.py's config.end_to_end reads), --gpus N / --devices A,B,.. (tile-sharded over
// the GPUs of one node, one RCCL gather of the float4 framebuffer at the end). Same flow: load scene -> convert through
// the add_* / set_* API -> init_bvh -> init_lights -> init_state -> sample loop
// -> save_image. Errors print and exit(1) like print_fatal
// (yocto_commonio.h:258-261).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <thread>

#include "yscene_cli.h"

int main(int argc, const char* argv[]) {
  auto params = ptr::trace_params{};
  auto save_batch = false, timing = false;
  std::string camera_name, imfilename = "out.hdr", filename, shader = "path";
  int  spp_per_launch = 256, gpus = 1, first_device = 0;  // (every launch waits for its unluckiest pixel: C1 0.2375 ms per sample at 64 per launch, 0.228 at 256, 0.2226 in one launch of 1536)
  std::string device_list;

  auto usage = [&]() {
    printf("usage: yscenetrace [--camera NAME] [--resolution,-r N] [--samples,-s N] [--shader,-t naive|path|eyelight|normal]\n"
           "                   [--bounces,-b N] [--clamp F] [--save-batch] [--output-image,-o FILE]\n"
           "                   [--seed N] [--exact-bsdf] [--device N] [--gpus N] [--devices A,B,..] [--spp-per-launch N] [--timing] scene\n"
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
    else if (a == "--timing") timing = true;
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
  yh_set_trial_cache_dir(yh_default_trial_cache_dir());  // the command line keeps its kernel-trial record on disk (include/yhair.h); a library caller has to ask
  set_devices(first_device, gpus, device_list);
  bool known = false;
  for (size_t i = 0; i < ptr::shader_names.size(); i++)
    if (ptr::shader_names[i] == shader) params.shader = (ptr::shader_type)i, known = true;
  if (!known) print_fatal("unknown shader " + shader);

  try {
    char error[512];
    auto t0 = std::chrono::steady_clock::now();
    auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    double t_last = 0, ph_load, ph_convert, ph_bvh, ph_lights, ph_state, ph_render, ph_download, ph_save;
    auto   lap = [&]() { double now = secs(), d = now - t_last; t_last = now; return d; };
    // the device contexts in a thread of their own while the scene file is read: creating the first one initialises the HIP runtime (a tenth of a second,
    // profiles/r06/e2e_laps_before.txt: it used to be the first thing init_state waited for); a failure is reported by the caller that needs them
    std::thread warm_contexts([] {
      try {
        (void)yhair::detail::require_contexts();
      } catch (...) {
      }
    });
    struct Joiner {
      std::thread& t;
      ~Joiner() { if (t.joinable()) t.join(); }
    } join_warm{warm_contexts};
    auto ioscene = yh_scene_load(filename.c_str(), camera_name.c_str(), error, sizeof(error));
    if (!ioscene) {
      warm_contexts.join();  // (print_fatal exits: never while the other thread is inside the HIP runtime's initialisation)
      print_fatal(error);
    }
    ph_load = lap();
    printf("load scene: %.2fs\n", ph_load);

    auto scene  = std::make_unique<ptr::scene>();
    auto camera = init_scene(scene.get(), yh_scene_get(ioscene));
    yh_scene_free(ioscene);
    ph_convert = lap();

    warm_contexts.join();
    ptr::init_bvh(scene.get(), params);  // flattens the scene graph and uploads it: yh_upload_scene = BVH build + records + copies
    ph_bvh = lap();
    ptr::init_lights(scene.get(), params);
    ph_lights = lap();
    auto state = std::make_unique<ptr::state>();
    ptr::init_state(state.get(), scene.get(), camera, params);  // pixel streams + the 1-spp probe launch that plans the first hand-out
    ph_state = lap();
    ph_bvh += scene->upload_seconds, ph_state -= scene->upload_seconds;  // (the mirror uploads in init_state, the first call that has the camera: yhair_pathtrace.h)
    printf("build bvh + lights + state (%dx%d): %.2fs\n", state->width, state->height, ph_convert + ph_bvh + ph_lights + ph_state);

    auto save = [&](const std::string& name) {
      if (yh_save_image(name.c_str(), state->width, state->height, (const float*)state->render.data(), error,
              sizeof(error)) != YH_OK)
        print_fatal(error);
    };
    double kernel_ms = 0;
    int    launches = 0, requests = 0;
    for (int sample = 0; sample < params.samples;) {
      int n = save_batch ? 1 : std::min(spp_per_launch, params.samples - sample);
      const double r0 = secs();
      ptr::trace_samples(state.get(), scene.get(), camera, params, n, save_batch);
      sample += n, requests++;
      float ms = 0;
      int   l  = 0;
      if (yh_last_trace_ms(yhair::detail::context(), &ms, &l) == YH_OK) kernel_ms += ms, launches += l;  // (context 0; the others run beside it)
      if (timing && getenv("YHAIR_TIMING")) fprintf(stderr, "[yhair] request %d: %d spp, %.2f ms wall, %.2f ms of kernels in %d launch(es)\n", requests, n, (secs() - r0) * 1e3, ms, l);
      if (save_batch) {  // cli.cpp:259-267: "<stem>-s<sample><ext>"
        auto dot_pos = imfilename.rfind('.');
        auto stem = imfilename.substr(0, dot_pos), ext = dot_pos == std::string::npos ? "" : imfilename.substr(dot_pos);
        save(stem + "-s" + std::to_string(sample - 1) + ext);
      }
    }
    ph_render = lap();
    ptr::trace_samples(state.get(), scene.get(), camera, params, 0, true);  // the image: one download, or the RCCL gather of the shards
    ph_download = lap();
    printf("render image: %d samples on %d GPU(s), %.3fs, %.1f Msamples/s\n", params.samples, (int)yhair::detail::devices().size(), ph_render + ph_download,
        (double)state->width * state->height * params.samples / (ph_render + ph_download) / 1e6);
    save(imfilename);
    ph_save = lap();
    printf("save image: %s\n", imfilename.c_str());
    if (timing)
      printf("timing: {\"load_scene_s\": %.4f, \"convert_s\": %.4f, \"bvh_and_upload_s\": %.4f, \"lights_s\": %.4f, \"init_state_and_probe_s\": %.4f, "
             "\"sample_loop_s\": %.4f, \"sample_loop_kernel_s\": %.4f, \"launches\": %d, \"requests\": %d, \"download_or_gather_s\": %.4f, \"save_s\": %.4f, "
             "\"total_s\": %.4f, \"width\": %d, \"height\": %d, \"samples\": %d, \"gpus\": %d}\n",
          ph_load, ph_convert, ph_bvh, ph_lights, ph_state, ph_render, kernel_ms / 1e3, launches, requests, ph_download, ph_save, secs(), state->width, state->height,
          params.samples, (int)yhair::detail::devices().size());
  } catch (const std::exception& e) {
    print_fatal(e.what());
  }
  return 0;
}
