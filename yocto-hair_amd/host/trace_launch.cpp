// trace_launch.cpp — yh_init_state (pt.cpp:1931-1946) and the launches of the sample-loop kernels: yh_trace_samples and friends.
#include "context_internal.h"

// Waits for everything queued on the context's stream, at most YHAIR_LAUNCH_TIMEOUT_S seconds (host/deadline.h): the blocking
// hipStreamSynchronize runs on the context's worker thread and this thread waits for it with the deadline, so that a kernel that never
// completes costs the caller an error, not the process. On expiry the context is POISONED: the device may still be running the kernel
// (and the worker is still inside the call), so every later call is refused and yh_destroy frees nothing (hipFree would wait too).
int wait_for_launch(yh_context* ctx) {
  if (ctx->poisoned) return fail(ctx, YH_E_DEVICE, "a launch of this context exceeded its deadline: the context refuses further work, destroy it");
  const double timeout = yhh::launch_timeout_s();
  const int    device  = ctx->device;
  hipStream_t  stream  = ctx->stream;
  int          e       = 0;
  // The blocking wait of the HIP runtime sleeps on an interrupt and wakes up tens of microseconds after the kernel has ended — for the bench's
  // 15 ms launches that was most of what the library added to a step (profiles/r05/bounded_wait_overhead.txt) — so the wait first SPINS for as
  // long as launches of this context have been taking (at most 50 ms, never past the deadline)
  // ... and for that long THIS thread asks the stream itself (hipStreamQuery: nothing blocks, nothing can hang): a launch of the usual length is
  // over before the question stops being asked, and the two thread hand-overs to the worker and back never happen.
  const double spin_s  = std::min(std::min(0.050, timeout), 1.25e-3 * (double)ctx->last_ms + 0.001);
  {
    const auto t0 = std::chrono::steady_clock::now();
    while (true) {
      const hipError_t q = hipStreamQuery(stream);
      if (q == hipSuccess) return YH_OK;
      if (q != hipErrorNotReady) return fail(ctx, YH_E_DEVICE, "hipStreamQuery: %s", hipGetErrorString(q));
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > spin_s) break;
    }
  }
  const int    rc      = ctx->sync_call.run(
      [device, stream]() {
        hipError_t se = hipSetDevice(device);
        return (int)(se != hipSuccess ? se : hipStreamSynchronize(stream));
      },
      timeout, &e);
  if (rc == yhh::WAIT_EXPIRED) {
    ctx->poisoned = true;
    return fail(ctx, YH_E_DEVICE, "the launch did not complete within %.3g s (YHAIR_LAUNCH_TIMEOUT_S): the context refuses further launches; destroy it and, to retry, start a fresh process", timeout);
  }
  if (e != (int)hipSuccess) return fail(ctx, YH_E_DEVICE, "hipStreamSynchronize: %s", hipGetErrorString((hipError_t)e));
  return YH_OK;
}

int yh_init_state(yh_context* ctx, const yh_trace_params* params) {
  if (!ctx) return YH_E_INVALID;
  if (ctx->poisoned) return fail(ctx, YH_E_DEVICE, "a launch of this context exceeded its deadline: the context refuses further work, destroy it");
  if (!ctx->have_scene) return fail(ctx, YH_E_STATE, "yh_init_state before yh_upload_scene");
  if (!params || params->resolution <= 0 || params->bounces < 0)
    return fail(ctx, YH_E_INVALID, "bad trace params");
  if (params->shader < 0 || params->shader >= YH_SHADER_COUNT)
    return fail(ctx, YH_E_INVALID, "sampler unknown");  // get_trace_shader_func's throw (pt.cpp:1669)
  if (params->hair_exact && params->shader != YH_SHADER_PATH) return fail(ctx, YH_E_INVALID, "hair_exact exists for the path shader only");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  YH_WAIT(ctx);  // (an asynchronous launch may still be running on the buffers this call replaces: wait for it, within the deadline)
  const bool timing = getenv("YHAIR_TIMING") && atoi(getenv("YHAIR_TIMING")) != 0;
  auto       t_last = std::chrono::steady_clock::now();
  auto       lap    = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[yhair] init_state: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  const bool same_work = ctx->have_state && ctx->params.shader == params->shader && ctx->params.bounces == params->bounces;  // (what the items' relative costs depend on besides the image)
  ctx->params = *params;
  // image size (pt.cpp:1933-1939)
  auto& cam = ctx->scene.camera;
  int   w, h;
  if (cam.film_x > cam.film_y) {
    w = params->resolution;
    h = (int)round(params->resolution * cam.film_y / cam.film_x);
  } else {
    w = (int)round(params->resolution * cam.film_x / cam.film_y);
    h = params->resolution;
  }
  if (w <= 0 || h <= 0) return fail(ctx, YH_E_INVALID, "empty image");
  size_t npix = (size_t)w * h;
  // per-pixel streams (pt.cpp:1942-1945), pixel order j * W + i
  std::vector<uint64_t> st(npix), inc(npix);
  Rng master = make_rng(1301081);
  for (size_t i = 0; i < npix; i++) {
    int seq = (int)(advance_rng(master) % 2147483648u) / 2 + 1;  // rand1i(rng, 1 << 31) / 2 + 1
    Rng r   = make_rng(params->seed, (uint64_t)seq);
    st[i] = r.state, inc[i] = r.inc;
  }
  lap("pixel streams (host)");
  int tx = tiles_of(w), ty = tiles_of(h);
  ctx->num_tiles_total = tx * ty;
  auto& owned = ctx->owned;
  owned.clear();
  for (int t = ctx->rank; t < ctx->num_tiles_total; t += ctx->world) owned.push_back(t);
  bool new_image = false;
  if ((int)ctx->item_cost.size() != ctx->num_tiles_total * 4) {  // scheduling hints survive a re-init of the same image
    ctx->item_cost.assign((size_t)ctx->num_tiles_total * 4, 0);
    ctx->item_scale.clear(), ctx->st_log_fresh = false;
    ctx->have_costs = false, ctx->costs_settled = false, ctx->dense = -1, ctx->chain = -1, ctx->chain16 = -1, ctx->launch_shape = 0;
    for (double& t : ctx->shape_ms) t = 0;
    for (int& t : ctx->shape_trials) t = 0;
    new_image = true;
  }
  std::vector<int> tiles;
  build_work_items(ctx, tiles);
  const int first_shape = getenv("YHAIR_SHAPE") ? choose_launch_shape(ctx) : ctx->launch_shape;
  ctx->state.tiles_x = tx, ctx->state.num_groups = 1, ctx->state.group_begin[0] = 0, ctx->state.group_begin[1] = (int)tiles.size();
  if (params->shader == YH_SHADER_PATH && first_shape == 3) deal_items_for_stream(ctx, tiles);
  if (params->shader == YH_SHADER_PATH && (first_shape == 4 || first_shape == 7)) split_items_for_octets(tiles);
  if (params->shader == YH_SHADER_PATH && first_shape == 5) split_items_side_by_side(ctx, tiles);
  if (params->shader == YH_SHADER_PATH && (first_shape == 6 || first_shape == 8)) split_items_for_hex(tiles);
  lay_out_first_round(ctx, tiles, params->shader == YH_SHADER_PATH ? first_shape : 0);
  tiles.reserve(4 * (size_t)ctx->num_tiles_total * 4 + 4);  // (the list's buffer holds the octet / sixteen-lane kernels' longer lists too)
  int rc;
  if ((rc = upload(ctx, ctx->d_rng_state, st.data(), npix * 8))) return rc;
  if ((rc = upload(ctx, ctx->d_rng_inc, inc.data(), npix * 8))) return rc;
  if ((rc = alloc_zero(ctx, ctx->d_accum, npix * 16))) return rc;
  if ((rc = alloc_zero(ctx, ctx->d_image, npix * 16))) return rc;
  {
    const size_t n = tiles.size();
    tiles.resize(std::max(n, 4 * owned.size() * 4), 0);
    if ((rc = upload(ctx, ctx->d_tiles, tiles.data(), tiles.size() * 4))) return rc;
    tiles.resize(n);
  }
  if ((rc = alloc_zero(ctx, ctx->d_counters, sizeof(yhd_counters)))) return rc;
  if ((rc = alloc_zero(ctx, ctx->d_tile_cursor, 8 * 16 * 4))) return rc;  // one cursor, or k_stream's one per item group 64 bytes apart
  if ((rc = alloc_zero(ctx, ctx->d_tile_cost, (size_t)ctx->num_tiles_total * 16))) return rc;
  auto& s = ctx->state;
  s.tile_cursor = (int*)ctx->d_tile_cursor.p, s.tile_cost = (unsigned int*)ctx->d_tile_cost.p;
  s.rng_state = (uint64_t*)ctx->d_rng_state.p, s.rng_inc = (uint64_t*)ctx->d_rng_inc.p;
  s.accum = (yhd_float4*)ctx->d_accum.p, s.tiles = (const int*)ctx->d_tiles.p;
  s.launch_shape = first_shape;
  s.num_tiles = (int)tiles.size(), s.width = w, s.height = h, s.tiles_x = tx;
  if (new_image || !same_work) ctx->launches_of_image = 0;
  s.samples_done = 0, s.bounces = params->bounces, s.clamp = params->clamp, s.shader = params->shader;
  s.shard_rank = ctx->rank, s.shard_world = ctx->world;
  ctx->have_state = true;
  lap("work list, buffers, copies");
  if (new_image) trials_load(ctx);  // what this process already measured on this scene, image and shard
  // Probe: the first launch of a new image has no item costs and would hand its work items out in image order,
  // 25-60 % slower than a planned launch (hair quadrants cost 10-100x background ones and bound the launch when
  // they start last). One sample of every pixel measures them; the state is then put back as it was, so the
  // render starts planned and from the reference's RNG states. (YHAIR_NO_PROBE: developer switch.)
  bool measured = false;
  for (int t : owned)
    for (int q = 0; q < 4 && !measured; q++) measured = ctx->item_cost[(size_t)t * 4 + q] != 0;
  if (!measured && !owned.empty() && params->shader == YH_SHADER_PATH && !getenv("YHAIR_NO_PROBE")) {
    ctx->state.launch_shape = 0;
    int prc = trace_impl(ctx, 1, false, true);  // blocking; re-plans the hand-out order from the measured costs
    if (prc) return prc;
    HIPCHK(ctx, hipMemcpy(ctx->d_rng_state.p, st.data(), npix * 8, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_accum.p, 0, npix * 16, ctx->stream));
    YH_WAIT(ctx);
    ctx->state.samples_done = 0, ctx->launches_of_image = 0, ctx->last_shape = -1, ctx->last_ms = 0, ctx->last_launches = 0;
    ctx->have_costs = true;  // the launches that follow are planned: their times rank the kernels
    lap("probe launch + plan");
  }
  return YH_OK;
}

// One launch of the streaming integrator (csrc/stream.hip): persistent wavefronts, one path pool each, one lane per path.
int stream_impl(yh_context* ctx, int nsamples, bool sync) {
  int  P = 0, grid = 0, lds_bytes = 0;
  bool one_generation = false;
  if (!lane_kernels_can_address(ctx))
    return fail(ctx, YH_E_INVALID, "k_stream (launch shape 3) reads the scene's trees through 32-bit byte offsets and this scene's exceed 4 GB: the quad kernels render it");
  if (!stream_geometry(ctx, ctx->st_items > 0 ? ctx->st_items : ctx->state.num_tiles, &P, &grid, &lds_bytes, &one_generation))
    return fail(ctx, YH_E_DEVICE, "k_stream cannot run with its LDS layout on this device");
  const int     wpb    = yhk_stream_block_threads() / 64;
  size_t        waves  = (size_t)grid * wpb;
  // The per-wave shares of the work list (deal_shares_by_speed) are made for ONE launch geometry — a wave count and a pool size — and the
  // cursor then starts behind the whole list: launched with another geometry (YHAIR_ST_WAVES / YHAIR_ST_SLOTS changed between the plan and
  // the launch), waves beyond the plan's would find nothing, shares beyond the pool's slots would be dropped, and samples_done would still
  // advance over pixels nobody rendered. So: a mismatch re-deals the list for the geometry of THIS launch; if that does not settle it, the
  // launch is refused — never a list the waves cannot reach.
  if (ctx->stream_pool.wave_begin && (ctx->st_share_waves != waves || ctx->st_share_slots != P)) {
    if (int wrc = wait_for_launch(ctx)) return wrc;
    if (int rc = upload_work_items(ctx)) return rc;
    if (!stream_geometry(ctx, ctx->st_items > 0 ? ctx->st_items : ctx->state.num_tiles, &P, &grid, &lds_bytes, &one_generation))
      return fail(ctx, YH_E_DEVICE, "k_stream cannot run with its LDS layout on this device");
    waves = (size_t)grid * wpb;
    if (ctx->stream_pool.wave_begin && (ctx->st_share_waves != waves || ctx->st_share_slots != P))
      return fail(ctx, YH_E_STATE, "k_stream: the work list was shared out for %zu waves of %d slots and the launch has %zu of %d", ctx->st_share_waves, ctx->st_share_slots, waves, P);
  }
  const size_t  slots  = waves * P;
  // overflow of the per-lane LDS stack windows (dev_lane.h): a main ray plus a light-pdf ray above it
  const int    ovf_entries = 2 * std::max(8, ctx->stack_need);
  const size_t ovf_words   = waves * (size_t)ovf_entries * 64;
  int rc;
  if (slots > ctx->st_slots) {
    static_assert(sizeof(yhd_path_slot) == 128, "a path slot is one cache line");
    if ((rc = alloc_zero(ctx, ctx->d_st_slots, slots * sizeof(yhd_path_slot)))) return rc;
    ctx->st_slots          = slots;
    ctx->stream_pool.slots = (yhd_path_slot*)ctx->d_st_slots.p;
  }
  if (ctx->scene.general_materials && slots > ctx->st_medium_slots) {
    if ((rc = alloc_zero(ctx, ctx->d_st_medium, slots * 32))) return rc;
    ctx->st_medium_slots = slots, ctx->stream_pool.medium = (yhd_float4*)ctx->d_st_medium.p;
  }
  if (ovf_words > ctx->st_ovf_words) {
    if ((rc = alloc_zero(ctx, ctx->d_st_ovf, ovf_words * 4))) return rc;
    ctx->st_ovf_words = ovf_words, ctx->stream_pool.stack_ovf = (unsigned int*)ctx->d_st_ovf.p;
  }
  ctx->stream_pool.slots_per_wave = P, ctx->stream_pool.ovf_entries = ovf_entries, ctx->stream_pool.total_slots = (long long)ctx->st_slots;
  ctx->stream_pool.suspend_lanes  = one_generation ? 8 : 16;  // (csrc/stream.hip: YH_SUSPEND_LANES)
  const bool prof = getenv("YHAIR_ST_PROF") && atoi(getenv("YHAIR_ST_PROF")) != 0;  // developer switch: per-stage counters on stderr
  if (prof) {
    if ((rc = alloc_zero(ctx, ctx->d_st_prof, 64 * 8))) return rc;
    ctx->stream_pool.prof = (unsigned long long*)ctx->d_st_prof.p;
    ctx->stream_pool.prof_parts_only = atoi(getenv("YHAIR_ST_PROF")) == 2;
  } else {
    ctx->stream_pool.prof = nullptr;
  }
  // every launch keeps the waves' begin / end stamps and work (32 bytes per wave): the speeds of the hardware wave slots the next
  // hand-out is sized by (launch_plan.cpp: note_stream_wave_log); YHAIR_ST_WAVELOG prints their distribution
  const bool wave_log = getenv("YHAIR_ST_WAVELOG") != nullptr;
  if (ctx->d_st_wave_log.bytes < waves * 16)
    if ((rc = alloc_zero(ctx, ctx->d_st_wave_log, waves * 16))) return rc;
  ctx->stream_pool.wave_log = (unsigned long long*)ctx->d_st_wave_log.p;
  ctx->stream_pool.wave_fill = nullptr;
  if (ctx->stream_pool.wave_begin) {
    if (ctx->d_st_wave_fill.bytes < waves * 4)
      if ((rc = alloc_zero(ctx, ctx->d_st_wave_fill, waves * 4))) return rc;
    ctx->stream_pool.wave_fill = (int*)ctx->d_st_wave_fill.p;
  }
  if (!ctx->d_scene_copy.p) {  // the scene table in device memory, for the kernel's out-of-line callees
    if ((rc = upload(ctx, ctx->d_scene_copy, &ctx->scene, sizeof(yhd_scene)))) return rc;
  }
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cursor.p, 0, 8 * 16 * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cost.p, 0, (size_t)ctx->num_tiles_total * 16, ctx->stream));
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int e = yhk_stream(&ctx->scene, (const yhd_scene*)ctx->d_scene_copy.p, &ctx->state, nsamples, &ctx->stream_pool, grid, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_stream launch: %s", hipGetErrorString((hipError_t)e));
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->state.samples_done += nsamples;
  ctx->last_launches = 1;
  if (sync) {
    if (int wrc = wait_for_launch(ctx)) return wrc;
    HIPCHK(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->ev0, ctx->ev1));
    std::vector<unsigned long long> w(waves * 2);
    HIPCHK(ctx, hipMemcpy(w.data(), ctx->d_st_wave_log.p, w.size() * 8, hipMemcpyDeviceToHost));
    note_stream_wave_log(ctx, w.data(), waves);
    if (wave_log) {
      unsigned long long t0 = ~0ull, t1 = 0;
      for (size_t i = 0; i < waves; i++) t0 = std::min(t0, w[2 * i]), t1 = std::max(t1, w[2 * i + 1]);
      const double span = (double)(t1 - t0);
      // by hardware wave slot (= dispatch round of the wave's workgroup: num_cus workgroups per round) and by XCC (workgroup % 8)
      double sum_slot[16] = {}, n_slot[16] = {}, sum_x[8] = {}, n_x[8] = {}, start_slot[16] = {};
      std::vector<double> ends;
      double resident = 0;
      for (size_t i = 0; i < waves; i++) {
        const size_t blk = i / wpb;
        const int    sl = (int)std::min<size_t>(15, blk / ctx->num_cus), x = (int)(blk % 8);
        const double e = (double)(w[2 * i + 1] - t0) / span, b = (double)(w[2 * i] - t0) / span;
        sum_slot[sl] += e, n_slot[sl] += 1, sum_x[x] += e, n_x[x] += 1, start_slot[sl] += b;
        ends.push_back(e), resident += e - b;
      }
      {  // do the same waves end late launch after launch? correlation of the end stamps with the previous launch's (same wave count)
        static std::vector<double> prev;
        if (prev.size() == ends.size()) {
          double ma = 0, mb = 0, saa = 0, sbb = 0, sab = 0;
          for (size_t i = 0; i < ends.size(); i++) ma += ends[i], mb += prev[i];
          ma /= ends.size(), mb /= ends.size();
          for (size_t i = 0; i < ends.size(); i++) saa += (ends[i] - ma) * (ends[i] - ma), sbb += (prev[i] - mb) * (prev[i] - mb), sab += (ends[i] - ma) * (prev[i] - mb);
          fprintf(stderr, "[yhair]   correlation of the waves' ends with the previous launch's: %.3f\n", sab / std::sqrt(std::max(1e-30, saa * sbb)));
        }
        prev = ends;
      }
      std::sort(ends.begin(), ends.end());
      fprintf(stderr, "[yhair] k_stream wave log: %zu waves, span %.2f ms; a wave is resident %.3f of the span on average; ends at p10 %.3f p50 %.3f p90 %.3f p99 %.3f\n", waves,
          span / 1e5, resident / waves, ends[waves / 10], ends[waves / 2], ends[waves * 9 / 10], ends[waves * 99 / 100]);
      for (int k = 0; k < 16; k++)
        if (n_slot[k] > 0) fprintf(stderr, "[yhair]   dispatch round %d: %4.0f waves, begin %.3f, mean end %.3f\n", k, n_slot[k], start_slot[k] / n_slot[k], sum_slot[k] / n_slot[k]);
      fprintf(stderr, "[yhair]   mean end by XCC:");
      for (int k = 0; k < 8; k++) fprintf(stderr, " %.3f", n_x[k] > 0 ? sum_x[k] / n_x[k] : 0.0);
      fprintf(stderr, "\n");
    }
    if (prof) {
      unsigned long long c[64];
      HIPCHK(ctx, hipMemcpy(c, ctx->d_st_prof.p, sizeof(c), hipMemcpyDeviceToHost));
      const char* names[6] = {"items", "sort", "finish", "hair", "surf", "trace"};
      double total = 0;
      for (int k = 0; k < 6; k++) total += (double)c[k];
      fprintf(stderr, "[yhair] k_stream %.2f ms, grid %d x %d waves, %d slots per wave\n", ctx->last_ms, grid, wpb, P);
      for (int k = 0; k < 6; k++)
        fprintf(stderr, "[yhair]   %-7s %5.1f %% of wave time, %9llu trips, mean batch %.1f lanes\n", names[k], 100.0 * (double)c[k] / total,
            c[8 + k], c[8 + k] ? (double)c[16 + k] / (double)c[8 + k] : 0.0);
      fprintf(stderr, "[yhair]   trace: %llu wave steps, %.1f lanes busy on average, %.0f cycles per step\n", c[24],
          c[24] ? (double)c[25] / (double)c[24] : 0.0, c[24] ? (double)c[5] / (double)c[24] : 0.0);
      {  // the five parts of a step (csrc/dev_lane.h: stamp), shader-clock cycles per wave step
        const char* part[5] = {"head (pop, scene level, ENTER)", "own loads, exchange, segment loads issued", "ray pulls, the wait for memory, node code", "the wave's line tests",
            "results back, accept"};
        double      sum = 0;
        for (int k = 0; k < 5; k++) sum += (double)c[52 + k];
        for (int k = 0; k < 5 && c[24]; k++)
          fprintf(stderr, "[yhair]   step part %d %-46s %7.0f cycles per step (%4.1f %% of the stamped step)\n", k, part[k], (double)c[52 + k] / (double)c[24], sum > 0 ? 100.0 * (double)c[52 + k] / sum : 0.0);
      }
      // per branch of lane_step (csrc/dev_lane.h: LP_*): the share of the wave steps that ran it, and the lanes in it when it ran
      const char* br[10] = {"step", "pop", "scene", "enter", "fetch", "node", "line-leaf", "tri-leaf", "push", "2nd-seg"};
      for (int b = 0; b < 10; b++)
        fprintf(stderr, "[yhair]   branch %-9s ran in %5.1f %% of the wave steps (%llu times), %.1f lanes on average\n", br[b],
            c[32] ? 100.0 * (double)c[32 + 2 * b] / (double)c[32] : 0.0, c[32 + 2 * b], c[32 + 2 * b] ? (double)c[33 + 2 * b] / (double)c[32 + 2 * b] : 0.0);
    }
    return replan_after_launch(ctx, nsamples);
  }
  return YH_OK;
}

int trace_impl(yh_context* ctx, int nsamples, bool counted, bool sync) {
  if (!ctx) return YH_E_INVALID;
  if (ctx->poisoned) return fail(ctx, YH_E_DEVICE, "a launch of this context exceeded its deadline: the context refuses further launches, destroy it");
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_trace_samples before yh_init_state");
  if (nsamples < 0) return fail(ctx, YH_E_INVALID, "negative sample count");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (nsamples == 0 || ctx->owned.empty()) {
    ctx->state.samples_done += nsamples;
    ctx->last_ms = 0, ctx->last_launches = 0;
    return YH_OK;
  }
  const bool path = ctx->state.shader == YH_SHADER_PATH;
  if (counted && !path) return fail(ctx, YH_E_INVALID, "work counters exist for the path shader only");
  if (path && !counted) {  // the kernel for this launch; the hand-out order follows it
    const int want = pick_launch_shape(ctx, sync ? nsamples : 0);  // (an asynchronous launch is not timed: never a trial)
    if (want != ctx->state.launch_shape) {
      if (getenv("YHAIR_TIMING"))
        fprintf(stderr, "[yhair] kernel times (ms per spp): 0: %.4f, 1: %.4f, 2: %.4f, 3: %.4f, 4: %.4f, 5: %.4f, 6: %.4f, 7: %.4f, 8: %.4f -> %d (%d spp)\n", ctx->shape_ms[0], ctx->shape_ms[1], ctx->shape_ms[2], ctx->shape_ms[3], ctx->shape_ms[4], ctx->shape_ms[5], ctx->shape_ms[6], ctx->shape_ms[7], ctx->shape_ms[8], want, nsamples);
      ctx->launch_shape = ctx->state.launch_shape = want;
      // The list is rewritten by a blocking copy on the null stream; the context's stream is non-blocking, so a launch
      // queued by yh_trace_samples_async may still be reading it: wait for it first.
      if (int wrc = wait_for_launch(ctx)) return wrc;
      if (int rc = upload_work_items(ctx)) return rc;
    }
  }
  if (counted && ctx->state.launch_shape == 3) {  // k_stream's list may be shared out per wave and padded with -1 (deal_shares_by_speed): the instrumented quad build needs a plain one
    ctx->launch_shape = ctx->state.launch_shape = ctx->dense > 0 ? 1 : 0;
    if (int wrc = wait_for_launch(ctx)) return wrc;
    if (int rc = upload_work_items(ctx)) return rc;
  }
  if (counted && (ctx->state.launch_shape == 5 || (ctx->state.launch_shape >= 4 && (ctx->scene.general_materials || ctx->state.launch_shape >= 7)))) {  // the octet kernel's list holds half-quadrant entries: the instrumented (quad) build needs its own
    ctx->launch_shape = ctx->state.launch_shape = 0;
    if (int wrc = wait_for_launch(ctx)) return wrc;
    if (int rc = upload_work_items(ctx)) return rc;
  }
  int shape = path ? ctx->state.launch_shape : 0;  // the preview shaders have one launch shape
  if (counted && (shape == 3 || (shape >= 2 && ctx->scene.general_materials))) shape = shape == 3 ? 1 : 0;  // no instrumented build of k_stream, nor of the GENERAL 8-wide forms
  if (shape == 3 && !getenv("YHAIR_SHAPE")) {      // a candidate that cannot run here is dropped, not an error: k_trace renders the same bits
    int P = 0, grid = 0;
    if (!stream_geometry(ctx, ctx->st_items > 0 ? ctx->st_items : ctx->state.num_tiles, &P, &grid, nullptr)) {
      ctx->shape_ms[3] = std::numeric_limits<double>::infinity();
      shape = ctx->dense > 0 ? 1 : 0;
      ctx->launch_shape = ctx->state.launch_shape = shape;
      if (int wrc = wait_for_launch(ctx)) return wrc;
      if (int rc = upload_work_items(ctx)) return rc;
    }
  }
  if ((shape == 4 || shape >= 6) && !counted && !getenv("YHAIR_SHAPE") &&
      yhk_trace_occupancy(yhk_trace_lds_bytes(&ctx->scene, shape), ctx->scene.general_materials, shape) < 1) {  // (likewise: a tree too deep for the wide forms' LDS stacks)
    ctx->shape_ms[shape] = std::numeric_limits<double>::infinity();
    shape = 0;
    ctx->launch_shape = ctx->state.launch_shape = shape;
    if (int wrc = wait_for_launch(ctx)) return wrc;
    if (int rc = upload_work_items(ctx)) return rc;
  }
  ctx->last_shape = shape, ctx->last_counted = counted, ctx->planned_settled = ctx->costs_settled;
  if (shape == 3) return stream_impl(ctx, nsamples, sync);
  if (shape == 2) return fail(ctx, YH_E_INVALID, "launch shape 2 (quads over 8-wide nodes) was a developer kernel and is not built (profiles/r03/w8_oct_ab.txt)");
  if (shape == 5 && !counted) return side_by_side_impl(ctx, nsamples, sync);
  if (shape == 5) shape = 0;  // (instrumented: guarded above, the list was rebuilt for the quad kernel)
  int waves_per_block = yhk_block_threads(shape) / 64;  // one work item per wave at a time
  int lds_bytes       = yhk_trace_lds_bytes(&ctx->scene, shape);
  const bool exact    = path && ctx->params.hair_exact && !counted;
  int occupancy       = exact ? yhk_trace_exact_occupancy(lds_bytes, ctx->scene.general_materials) : yhk_trace_occupancy(lds_bytes, ctx->scene.general_materials, shape);
  if (occupancy < 1) return fail(ctx, YH_E_DEVICE, "k_trace cannot run with %d bytes of LDS per block", lds_bytes);
  int resident        = ctx->num_cus * occupancy;
  int want            = (ctx->state.num_tiles + waves_per_block - 1) / waves_per_block;
  int grid            = std::max(1, std::min(want, resident));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cursor.p, 0, 8 * 16 * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cost.p, 0, (size_t)ctx->num_tiles_total * 16, ctx->stream));
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int e = exact ? yhk_trace_exact(&ctx->scene, &ctx->state, nsamples, lds_bytes, grid, ctx->stream)
                : yhk_trace(&ctx->scene, &ctx->state, nsamples, counted ? (yhd_counters*)ctx->d_counters.p : nullptr, shape, grid, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_trace launch: %s", hipGetErrorString((hipError_t)e));
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->state.samples_done += nsamples;
  ctx->last_launches = 1;
  if (sync) {
    if (int wrc = wait_for_launch(ctx)) return wrc;
    HIPCHK(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->ev0, ctx->ev1));
    return replan_after_launch(ctx, nsamples);
  }
  return YH_OK;
}
// One side-by-side launch: k_trace_sbs over the whole list — its first G_o workgroups the octet entries behind the quad items,
// the other G_q the quad items [0, hy_quad_items).
int side_by_side_impl(yh_context* ctx, int nsamples, bool sync) {
  int G_o = 0, G_q = 0;
  if (!side_by_side_grids(ctx, &G_o, &G_q)) return fail(ctx, YH_E_DEVICE, "k_trace_sbs cannot run with %d bytes of LDS per block", yhk_trace_sbs_lds_bytes(&ctx->scene));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cursor.p, 0, 8 * 16 * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cost.p, 0, (size_t)ctx->num_tiles_total * 16, ctx->stream));
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int e = yhk_trace_sbs(&ctx->scene, &ctx->state, nsamples, G_o, ctx->hy_quad_items, ctx->hy_oct_entries, std::max(1, G_o + G_q), ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_trace_sbs launch: %s", hipGetErrorString((hipError_t)e));
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->state.samples_done += nsamples;
  ctx->last_launches = 1;
  if (sync) {
    if (int wrc = wait_for_launch(ctx)) return wrc;
    HIPCHK(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->ev0, ctx->ev1));
    return replan_after_launch(ctx, nsamples);
  }
  return YH_OK;
}
int yh_trace_samples(yh_context* ctx, int nsamples) {
  if (!ctx) return YH_E_INVALID;
  // a long request starts with the short trial launches of the kernels this image has not timed yet (pick_launch_shape)
  float ms = 0;
  int   launches = 0, remaining = nsamples;
  do {
    const int n  = (remaining >= 2 * YH_TRIAL_SPP && trial_pending(ctx)) ? YH_TRIAL_SPP : remaining;
    const int rc = trace_impl(ctx, n, false, true);
    if (rc) return rc;
    ms += ctx->last_ms, launches += ctx->last_launches, remaining -= n;
  } while (remaining > 0);
  ctx->last_ms = ms, ctx->last_launches = launches;
  return YH_OK;
}
int yh_trace_samples_async(yh_context* ctx, int nsamples) {
  const int rc = trace_impl(ctx, nsamples, false, false);
  if (ctx && rc == YH_OK) ctx->async_pending = ctx->last_launches > 0;
  return rc;
}
int yh_synchronize(yh_context* ctx) {
  if (!ctx) return YH_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (int wrc = wait_for_launch(ctx)) return wrc;
  if (ctx->async_pending) (void)hipEventElapsedTime(&ctx->last_ms, ctx->ev0, ctx->ev1);  // (a blocking call has its own sum)
  ctx->async_pending = false;
  return YH_OK;
}
int yh_launch_shape(const yh_context* ctx) { return ctx ? ctx->last_shape : YH_E_INVALID; }
int yh_kernel_trials(const yh_context* ctx, double* ms_per_sample, int* trials, int count) {
  if (!ctx || !ms_per_sample || !trials || count < 1) return YH_E_INVALID;
  for (int k = 0; k < count; k++) {
    ms_per_sample[k] = k < YH_SHAPES ? ctx->shape_ms[k] : 0.0;
    trials[k]        = k < YH_SHAPES ? ctx->shape_trials[k] : 0;
    if (std::isinf(ms_per_sample[k])) ms_per_sample[k] = -1.0;  // a candidate that cannot run on this device
  }
  return YH_SHAPES;
}
int yh_trials_pending(const yh_context* ctx) { return ctx ? (trial_pending(ctx) ? 1 : 0) : YH_E_INVALID; }
int yh_last_trace_ms(const yh_context* ctx, float* ms, int* launches) {
  if (!ctx) return YH_E_INVALID;
  if (ms) *ms = ctx->last_ms;
  if (launches) *launches = ctx->last_launches;
  return YH_OK;
}
int yh_trace_samples_counted(yh_context* ctx, int nsamples, yh_workcounts* out) {
  if (!ctx || !out) return YH_E_INVALID;
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_trace_samples_counted before yh_init_state");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_counters.p, 0, sizeof(yhd_counters), ctx->stream));
  int rc = trace_impl(ctx, nsamples, true, true);
  if (rc) return rc;
  yhd_counters c;
  HIPCHK(ctx, hipMemcpy(&c, ctx->d_counters.p, sizeof(c), hipMemcpyDeviceToHost));
  out->samples = c.samples, out->rays = c.rays, out->nodes = c.nodes, out->seg_tests = c.seg, out->tri_tests = c.tri;
  out->hair_shades = c.hair, out->surf_shades = c.surf, out->env_lookups = c.envl, out->env_samples = c.envs;
  out->cyc_trace = c.cyc_trace, out->cyc_shade = c.cyc_shade, out->ticks_tile = c.cyc_tile, out->wave_iters = c.wave_iters;
  out->wave_steps = c.wave_steps, out->lane_steps = c.lane_steps, out->lane_iters = c.lane_iters;
  out->cyc_geom = c.c_geom, out->cyc_sample = c.c_sample, out->cyc_eval = c.c_eval, out->cyc_rest = c.c_rest;
  for (int k = 0; k < 10; k++) out->branch[k] = c.branch[k];
  return YH_OK;
}
