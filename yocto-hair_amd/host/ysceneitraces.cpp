// ysceneitraces — the reference's progressive path tracer (apps/ysceneitraces/ysceneitraces.cpp) without its
// window: the OTHER caller of trace_samples, through the same C++ mirror (yhair_pathtrace.h).
//
// What it restates is reset_display (ysceneitraces.cpp:255-300), the only part of the app that touches the
// path: init_state of the render state; a PREVIEW state at resolution / pratio traced for one sample and
// upscaled into the display image; then an asynchronous worker that calls the stop-flag overload of
// trace_samples once per sample (pt.cpp:2009-2026) until the samples are done or the flag is set — which is
// what a camera edit does in the app and what --stop-after-ms does here.
//
// Same flags as the app (ysceneitraces.cpp:313-327): --camera, --resolution,-r, --samples,-s, --shader,-t,
// --bounces,-b, --clamp, --output,-o, positional scene. Extensions: --pratio (trace_params::pratio, default
// 8), --preview-image FILE, --stop-after-ms N, --seed, --device, --gpus / --devices.
#include <atomic>
#include <chrono>
#include <future>
#include <thread>

#include "yscene_cli.h"

int main(int argc, const char* argv[]) {
  auto        params = ptr::trace_params{};
  std::string camera_name, imagename = "out.hdr", preview_name, filename, shader = "path";
  int         stop_after_ms = -1, gpus = 1, first_device = 0;
  std::string device_list;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> std::string {
      if (i + 1 >= argc) print_fatal("missing value for " + a);
      return argv[++i];
    };
    if (a == "--help" || a == "-h") {
      printf("usage: ysceneitraces [--camera NAME] [--resolution,-r N] [--samples,-s N] [--shader,-t naive|path|eyelight|normal]\n"
             "                     [--bounces,-b N] [--clamp F] [--output,-o FILE] [--pratio N] [--preview-image FILE]\n"
             "                     [--stop-after-ms N] [--seed N] [--device N] [--gpus N] [--devices A,B,..] scene\n"
             "Progressive path tracing of hair scenes on MI355X (headless: preview pass, then samples until done or stopped)\n");
      return 0;
    } else if (a == "--camera") camera_name = next();
    else if (a == "--resolution" || a == "-r") params.resolution = atoi(next().c_str());
    else if (a == "--samples" || a == "-s") params.samples = atoi(next().c_str());
    else if (a == "--shader" || a == "-t") shader = next();
    else if (a == "--bounces" || a == "-b") params.bounces = atoi(next().c_str());
    else if (a == "--clamp") params.clamp = (float)atof(next().c_str());
    else if (a == "--output" || a == "-o") imagename = next();
    else if (a == "--pratio") params.pratio = std::max(1, atoi(next().c_str()));
    else if (a == "--preview-image") preview_name = next();
    else if (a == "--stop-after-ms") stop_after_ms = atoi(next().c_str());
    else if (a == "--seed") params.seed = strtoull(next().c_str(), nullptr, 10);
    else if (a == "--device") first_device = atoi(next().c_str());
    else if (a == "--gpus") gpus = std::max(1, atoi(next().c_str()));
    else if (a == "--devices") device_list = next();
    else if (!a.empty() && a[0] == '-') print_fatal("unknown option " + a);
    else filename = a;
  }
  if (filename.empty()) print_fatal("missing scene");
  yh_set_trial_cache_dir(yh_default_trial_cache_dir());  // the command line keeps its kernel-trial record on disk (include/yhair.h); a library caller has to ask
  set_devices(first_device, gpus, device_list);
  bool known = false;
  for (size_t i = 0; i < ptr::shader_names.size(); i++)
    if (ptr::shader_names[i] == shader) params.shader = (ptr::shader_type)i, known = true;
  if (!known) print_fatal("unknown shader " + shader);

  try {
    char error[512];
    auto ioscene = yh_scene_load(filename.c_str(), camera_name.c_str(), error, sizeof(error));
    if (!ioscene) print_fatal(error);
    auto scene  = std::make_unique<ptr::scene>();
    auto camera = init_scene(scene.get(), yh_scene_get(ioscene));
    yh_scene_free(ioscene);
    ptr::init_bvh(scene.get(), params);
    ptr::init_lights(scene.get(), params);

    // ---- reset_display (ysceneitraces.cpp:255-300) -------------------------------------------------------
    auto render_state = std::make_unique<ptr::state>();
    ptr::init_state(render_state.get(), scene.get(), camera, params);
    const int          W = render_state->width, H = render_state->height;
    std::vector<vec4f> render((size_t)W * H);  // app->render: what the window shows
    // render preview
    auto t0     = std::chrono::steady_clock::now();
    auto pstate = std::make_unique<ptr::state>();
    auto pprms  = params;
    pprms.resolution /= params.pratio;
    pprms.samples = 1;
    ptr::init_state(pstate.get(), scene.get(), camera, pprms);
    ptr::trace_samples(pstate.get(), scene.get(), camera, pprms);
    for (int j = 0; j < H; j++)
      for (int i = 0; i < W; i++) {
        int pi = std::min(std::max(i / params.pratio, 0), pstate->width - 1), pj = std::min(std::max(j / params.pratio, 0), pstate->height - 1);
        render[(size_t)j * W + i] = pstate->render[(size_t)pj * pstate->width + pi];
      }
    printf("preview: %dx%d at 1 spp upscaled to %dx%d, %.1f ms\n", pstate->width, pstate->height, W, H,
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (!preview_name.empty() && yh_save_image(preview_name.c_str(), W, H, (const float*)render.data(), error, sizeof(error)) != YH_OK)
      print_fatal(error);
    // start render
    std::atomic<bool> render_stop{false};
    std::atomic<int>  render_counter{0};
    auto t1            = std::chrono::steady_clock::now();
    auto render_worker = std::async(std::launch::async, [&]() {
      for (int sample = 0; sample < params.samples; sample++) {
        ptr::trace_samples(render_state.get(), scene.get(), camera, params, &render_stop);
        if (render_stop) return;
        render = render_state->render;
        render_counter = sample + 1;
      }
    });
    if (stop_after_ms >= 0) {  // the user moves the camera: reset_display sets the flag and waits for the worker
      std::this_thread::sleep_for(std::chrono::milliseconds(stop_after_ms));
      render_stop = true;
    }
    render_worker.get();
    auto dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    printf("render: %d of %d samples in %.3fs%s\n", (int)render_counter, params.samples, dt, render_stop ? " (stopped)" : "");
    if (yh_save_image(imagename.c_str(), W, H, (const float*)render.data(), error, sizeof(error)) != YH_OK) print_fatal(error);
    printf("save image: %s\n", imagename.c_str());
  } catch (const std::exception& e) {
    print_fatal(e.what());
  }
  return 0;
}
