// context.cpp — the C ABI of include/yhair.h, part 1: the context (create / destroy / errors), shards and downloads; the shared
// helpers (fail, upload, alloc_zero). There is no CPU fallback: without a GPU yh_create returns NULL. See context_internal.h
// for the other translation units.
#include "context_internal.h"

std::string g_create_error = "no error";

int fail(yh_context* ctx, int code, const char* fmt, ...) {
  char    buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->error = buf;
  else g_create_error = buf;
  return code;
}

// (Both start with the BOUNDED wait for whatever the context has queued: freeing the old buffer is a device-synchronising hipFree and the copy a
// blocking one on the null stream — behind a kernel of yh_trace_samples_async that never completes they would wait for ever, outside the
// deadline include/yhair.h promises for every call that waits for the device.)
int upload(yh_context* ctx, DevBuf& buf, const void* src, size_t bytes) {
  YH_WAIT(ctx);
  buf.reset();
  size_t alloc = std::max<size_t>(bytes, 16);
  HIPCHK(ctx, hipMalloc(&buf.p, alloc));
  buf.bytes = alloc;
  if (bytes) HIPCHK(ctx, hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice));
  return YH_OK;
}
// The same into a buffer that is kept while it is large enough (no hipFree + hipMalloc per call: deal_shares_by_speed re-plans after every launch
// of a dense image).
int upload_keep(yh_context* ctx, DevBuf& buf, const void* src, size_t bytes) {
  YH_WAIT(ctx);
  if (!buf.p || buf.bytes < bytes) {
    buf.reset();
    const size_t alloc = std::max<size_t>(bytes + bytes / 4, 16);  // (room to grow: the list's length changes a little from plan to plan)
    HIPCHK(ctx, hipMalloc(&buf.p, alloc));
    buf.bytes = alloc;
  }
  if (bytes) HIPCHK(ctx, hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice));
  return YH_OK;
}
int alloc_zero(yh_context* ctx, DevBuf& buf, size_t bytes) {
  YH_WAIT(ctx);
  buf.reset();
  size_t alloc = std::max<size_t>(bytes, 16);
  HIPCHK(ctx, hipMalloc(&buf.p, alloc));
  buf.bytes = alloc;
  // The context's stream is non-blocking: a memset on the null stream would not be
  // ordered against kernels launched on it. Clear on that stream and wait, so the
  // buffer is zero for whoever touches it next (stream kernel or blocking copy).
  HIPCHK(ctx, hipMemsetAsync(buf.p, 0, alloc, ctx->stream));
  YH_WAIT(ctx);
  return YH_OK;
}

yhd_float4 node_lo(const yhh::Node& n) {
  yhd_float4 r{n.bbox.min[0], n.bbox.min[1], n.bbox.min[2], 0};
  memcpy(&r.w, &n.start, 4);
  return r;
}
yhd_float4 node_hi(const yhh::Node& n) {
  yhd_float4 r{n.bbox.max[0], n.bbox.max[1], n.bbox.max[2], 0};
  int        meta = ((int)(unsigned short)n.num) | ((int)n.internal << 16) | ((int)n.axis << 24);
  memcpy(&r.w, &meta, 4);
  return r;
}


const char* yh_version(void) { return "yhair 0.1 (gfx950, HIP)"; }

yh_context* yh_create(int device) {
  int        count = 0;
  hipError_t e     = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    fail(nullptr, YH_E_DEVICE, "no HIP device available (%s): the hair path has no CPU fallback",
        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0 || device >= count) {
    fail(nullptr, YH_E_INVALID, "device %d out of range (%d devices)", device, count);
    return nullptr;
  }
  auto ctx    = new yh_context{};
  ctx->device = device;
  hipDeviceProp_t prop;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess) {
    fail(nullptr, YH_E_DEVICE, "device %d setup failed: %s", device, hipGetErrorString(e));
    delete ctx;
    return nullptr;
  }
  ctx->num_cus = prop.multiProcessorCount;
  ctx->device_name = std::string(prop.gcnArchName) + "/" + prop.name + "/" + std::to_string(prop.multiProcessorCount);
  for (char& c : ctx->device_name)
    if (c == ' ' || c == '|' || c == '\n') c = '_';
  return ctx;
}

void yh_destroy(yh_context* ctx) {
  if (!ctx) return;
  if (ctx->poisoned) return;  // a launch exceeded its deadline (wait_for_launch): the device may still be running it — synchronising or freeing would wait for it; the process is expected to end
  destroy_communicators(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream && wait_for_launch(ctx) != YH_OK) return;  // (a queued launch that never completes: as above)
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* yh_last_error(const yh_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int yh_set_shard(yh_context* ctx, int rank, int world) {
  if (!ctx) return YH_E_INVALID;
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, YH_E_INVALID, "bad shard %d of %d", rank, world);
  if (rank != ctx->rank || world != ctx->world) ctx->item_cost.clear();  // another shard is another image to plan and to time kernels on: yh_init_state starts over
  ctx->rank = rank, ctx->world = world;
  ctx->have_state = false;
  return YH_OK;
}

int yh_image_size(const yh_context* ctx, int* width, int* height) {
  if (!ctx || !ctx->have_state) return YH_E_STATE;
  if (width) *width = ctx->state.width;
  if (height) *height = ctx->state.height;
  return YH_OK;
}

// The hand-out order of the work items for the kernel in ctx->state.launch_shape, from the item costs the host holds
// (it depends on the kernel: k_stream's items are dealt, not queued).
int yh_download(yh_context* ctx, float* rgba) {
  if (!ctx || !rgba) return YH_E_INVALID;
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_download before yh_init_state");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t bytes = (size_t)ctx->state.width * ctx->state.height * 16;
  HIPCHK(ctx, hipMemsetAsync(ctx->d_image.p, 0, bytes, ctx->stream));
  int e = yhk_resolve(&ctx->state, (int)ctx->owned.size(), ctx->state.samples_done, ctx->d_image.p, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_resolve launch: %s", hipGetErrorString((hipError_t)e));
  HIPCHK(ctx, hipMemcpyAsync(rgba, ctx->d_image.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  YH_WAIT(ctx);
  return YH_OK;
}

int yh_download_rng(yh_context* ctx, uint64_t* state_inc) {
  if (!ctx || !state_inc) return YH_E_INVALID;
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_download_rng before yh_init_state");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t npix = (size_t)ctx->state.width * ctx->state.height;
  std::vector<uint64_t> st(npix), inc(npix);
  YH_WAIT(ctx);
  HIPCHK(ctx, hipMemcpy(st.data(), ctx->d_rng_state.p, npix * 8, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(inc.data(), ctx->d_rng_inc.p, npix * 8, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < npix; i++) state_inc[2 * i] = st[i], state_inc[2 * i + 1] = inc[i];
  return YH_OK;
}

int yh_tile_costs(yh_context* ctx, uint32_t* ticks, int count) {
  if (!ctx || !ticks) return YH_E_INVALID;
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_tile_costs before yh_init_state");
  if (count < ctx->num_tiles_total) return fail(ctx, YH_E_INVALID, "buffer holds %d tiles, image has %d", count, ctx->num_tiles_total);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  YH_WAIT(ctx);
  std::vector<unsigned int> cost((size_t)ctx->num_tiles_total * 4);
  HIPCHK(ctx, hipMemcpy(cost.data(), ctx->d_tile_cost.p, cost.size() * 4, hipMemcpyDeviceToHost));
  for (int t = 0; t < ctx->num_tiles_total; t++)  // a tile is four work items (4x4 quadrants): report their sum
    ticks[t] = cost[4 * t] + cost[4 * t + 1] + cost[4 * t + 2] + cost[4 * t + 3];
  return YH_OK;
}

int yh_item_costs(yh_context* ctx, uint32_t* costs, int count) {
  if (!ctx || !costs) return YH_E_INVALID;
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_item_costs before yh_init_state");
  if (count < ctx->num_tiles_total * 4) return fail(ctx, YH_E_INVALID, "buffer holds %d items, image has %d", count, ctx->num_tiles_total * 4);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  YH_WAIT(ctx);
  HIPCHK(ctx, hipMemcpy(costs, ctx->d_tile_cost.p, (size_t)ctx->num_tiles_total * 16, hipMemcpyDeviceToHost));
  return YH_OK;
}
