// gather.cpp — tile packing and the ONE collective of the path: the framebuffer gather.
#include "context_internal.h"

int64_t yh_shard_pixels(const yh_context* ctx, int rank, int world) {
  if (!ctx || !ctx->have_state || world < 1 || rank < 0 || rank >= world) return -1;
  int64_t n = ctx->num_tiles_total > rank ? (ctx->num_tiles_total - rank + world - 1) / world : 0;
  return n * 64;
}
int yh_pack_tiles_device(yh_context* ctx, void* device_rgba, int64_t capacity, int64_t* count) {
  if (!ctx || !device_rgba) return YH_E_INVALID;
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_pack_tiles_device before yh_init_state");
  int64_t need = (int64_t)ctx->owned.size() * 64;
  if (capacity < need) return fail(ctx, YH_E_INVALID, "pack buffer too small (%lld < %lld pixels)", (long long)capacity, (long long)need);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int e = yhk_pack(&ctx->state, (int)ctx->owned.size(), ctx->state.samples_done, device_rgba, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_pack launch: %s", hipGetErrorString((hipError_t)e));
  YH_WAIT(ctx);
  if (count) *count = need;
  return YH_OK;
}
int yh_unpack_tiles_device(yh_context* ctx, const void* device_packed, int src_rank, int world, void* device_image) {
  if (!ctx || !device_packed || !device_image) return YH_E_INVALID;
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_unpack_tiles_device before yh_init_state");
  if (world < 1 || src_rank < 0 || src_rank >= world) return fail(ctx, YH_E_INVALID, "bad shard %d of %d", src_rank, world);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int n = (int)(yh_shard_pixels(ctx, src_rank, world) / 64);
  int e = yhk_unpack(device_packed, src_rank, world, n, ctx->num_tiles_total, ctx->state.tiles_x, ctx->state.width,
      ctx->state.height, device_image, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_unpack launch: %s", hipGetErrorString((hipError_t)e));
  YH_WAIT(ctx);
  return YH_OK;
}

namespace {
// librccl, opened on first use: libyhair.so itself does not link it (a one-GPU user never needs it, and under
// PyTorch the process already holds a librccl of its own that a second copy must not shadow)
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string open_error = "missing symbols";  // why rccl_api() returned NULL (dlerror() read once)
};
RcclApi* rccl_api(const char** why = nullptr) {
  static RcclApi        api;
  static std::once_flag once;  // the C++ mirror drives contexts from several host threads
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      api.CommInitAll    = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
      api.CommDestroy    = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
      api.CommCount      = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
      api.CommUserRank   = (decltype(api.CommUserRank))dlsym(api.lib, "ncclCommUserRank");
      api.GroupStart     = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
      api.GroupEnd       = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
      api.Gather         = (decltype(api.Gather))dlsym(api.lib, "ncclGather");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
      if (!api.CommInitAll || !api.CommDestroy || !api.CommCount || !api.CommUserRank || !api.GroupStart || !api.GroupEnd || !api.Gather || !api.GetErrorString)
        api.lib = nullptr;
    } else if (const char* why = dlerror()) {
      api.open_error = why;
    }
  });
  if (!api.lib && why) *why = api.open_error.c_str();
  return api.lib ? &api : nullptr;
}
}  // namespace

void destroy_communicators(yh_context* ctx) {
  if (ctx->comms.empty()) return;
  if (RcclApi* api = rccl_api())
    for (ncclComm_t c : ctx->comms)
      if (c) (void)api->CommDestroy(c);
  ctx->comms.clear(), ctx->comm_devices.clear();
}

int yh_gather_framebuffer(yh_context** ctxs, int n, float* rgba) {
  if (!ctxs || n < 1 || !rgba || !ctxs[0]) return YH_E_INVALID;
  yh_context* root = ctxs[0];
  for (int i = 0; i < n; i++) {
    yh_context* c = ctxs[i];
    if (!c) return fail(root, YH_E_INVALID, "context %d is NULL", i);
    if (!c->have_state) return fail(root, YH_E_STATE, "yh_gather_framebuffer: context %d has no state", i);
    if (c->rank != i || c->world != n) return fail(root, YH_E_INVALID, "context %d holds shard %d of %d, expected %d of %d", i, c->rank, c->world, i, n);
    if (c->state.width != root->state.width || c->state.height != root->state.height || c->state.samples_done != root->state.samples_done)
      return fail(root, YH_E_INVALID, "context %d renders a different image or sample count than context 0", i);
  }
  int64_t cap = 0;  // float4 pixels of the largest shard: ncclGather moves equal counts
  for (int i = 0; i < n; i++) cap = std::max<int64_t>(cap, yh_shard_pixels(root, i, n));
  const size_t cap_bytes = (size_t)std::max<int64_t>(cap, 1) * 16;
  bool distinct = true;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) distinct = distinct && ctxs[i]->device != ctxs[j]->device;
  // YHAIR_GATHER=peer: device-to-device copies even between distinct devices; YHAIR_GATHER=rccl: the collective even
  // for ONE context (a communicator of one rank: how a one-GPU box executes the RCCL calls, tests/test_gpu_parity.py)
  const char* mode_env = getenv("YHAIR_GATHER");
  const bool  force_rccl = mode_env && !strcmp(mode_env, "rccl");
  const bool  use_rccl = distinct && (n > 1 || force_rccl) && !(mode_env && !strcmp(mode_env, "peer"));
  // every context packs its own tiles on its own stream
  for (int i = 0; i < n; i++) {
    yh_context* c = ctxs[i];
    HIPCHK(root, hipSetDevice(c->device));
    if (c->d_gather_send.bytes < cap_bytes) {
      int rc = alloc_zero(c, c->d_gather_send, cap_bytes);
      if (rc) return fail(root, rc, "context %d: %s", i, c->error.c_str());
    }
    int e = yhk_pack(&c->state, (int)c->owned.size(), c->state.samples_done, c->d_gather_send.p, c->stream);
    if (e) return fail(root, YH_E_DEVICE, "k_pack launch on context %d: %s", i, hipGetErrorString((hipError_t)e));
  }
  HIPCHK(root, hipSetDevice(root->device));
  if (root->d_gather_recv.bytes < cap_bytes * n) {
    int rc = alloc_zero(root, root->d_gather_recv, cap_bytes * n);
    if (rc) return rc;
  }
  if (use_rccl) {
    const char* why = "";
    RcclApi*    api = rccl_api(&why);
    if (!api) return fail(root, YH_E_DEVICE, "yh_gather_framebuffer: librccl could not be opened (%s)", why);
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) devs[i] = ctxs[i]->device;
    if (root->comm_devices != devs) {  // communicators are made once per device set
      for (ncclComm_t c : root->comms) (void)api->CommDestroy(c);
      root->comms.assign(n, nullptr), root->comm_devices.clear();
      ncclResult_t r = api->CommInitAll(root->comms.data(), n, devs.data());
      if (r != ncclSuccess) {
        root->comms.clear();
        return fail(root, YH_E_DEVICE, "ncclCommInitAll: %s", api->GetErrorString(r));
      }
      // what RCCL made must be what was asked for: n ranks, communicator i = rank i (the gather's root is rank 0 and
      // un-interleaves shard r from the r-th block of the receive buffer)
      for (int i = 0; i < n; i++) {
        int count = -1, urank = -1;
        ncclResult_t rc1 = api->CommCount(root->comms[i], &count), rc2 = api->CommUserRank(root->comms[i], &urank);
        if (rc1 != ncclSuccess || rc2 != ncclSuccess || count != n || urank != i) {
          destroy_communicators(root);
          return fail(root, YH_E_DEVICE, "ncclCommInitAll made communicator %d with %d ranks as rank %d (wanted %d ranks, rank %d)", i, count, urank, n, i);
        }
      }
      root->comm_devices = devs;
    }
    ncclResult_t r = api->GroupStart();
    hipError_t   he = hipSuccess;  // the group is closed whatever happens inside it: an open group would hang the
                                   // process's next RCCL call (PyTorch's included)
    if (r == ncclSuccess) {
      for (int i = 0; i < n && r == ncclSuccess && he == hipSuccess; i++) {
        if ((he = hipSetDevice(ctxs[i]->device)) != hipSuccess) break;
        r = api->Gather(ctxs[i]->d_gather_send.p, i == 0 ? root->d_gather_recv.p : nullptr, (size_t)cap * 4, ncclFloat, 0, root->comms[i], ctxs[i]->stream);
      }
      ncclResult_t r2 = api->GroupEnd();
      if (r == ncclSuccess) r = r2;
    }
    if (he != hipSuccess) return fail(root, YH_E_DEVICE, "yh_gather_framebuffer: hipSetDevice: %s", hipGetErrorString(he));
    if (r != ncclSuccess) return fail(root, YH_E_DEVICE, "ncclGather: %s", api->GetErrorString(r));
    for (int i = 0; i < n; i++) {
      HIPCHK(root, hipSetDevice(ctxs[i]->device));
      if (int wrc = wait_for_launch(ctxs[i])) return ctxs[i] == root ? wrc : fail(root, wrc, "yh_gather_framebuffer: context %d: %s", i, ctxs[i]->error.c_str());
    }
  } else {
    for (int i = 0; i < n; i++) {  // device-to-device copies (contexts sharing a device, or YHAIR_GATHER=peer)
      yh_context* c = ctxs[i];
      HIPCHK(root, hipSetDevice(c->device));
      if (int wrc = wait_for_launch(c)) return c == root ? wrc : fail(root, wrc, "yh_gather_framebuffer: context %d: %s", i, c->error.c_str());
      HIPCHK(root, hipSetDevice(root->device));
      void* dst = (char*)root->d_gather_recv.p + (size_t)i * cap_bytes;
      if (c->device == root->device) HIPCHK(root, hipMemcpyAsync(dst, c->d_gather_send.p, cap_bytes, hipMemcpyDeviceToDevice, root->stream));
      else HIPCHK(root, hipMemcpyPeerAsync(dst, root->device, c->d_gather_send.p, c->device, cap_bytes, root->stream));
    }
  }
  // the root un-interleaves every shard into the full image
  HIPCHK(root, hipSetDevice(root->device));
  const size_t bytes = (size_t)root->state.width * root->state.height * 16;
  HIPCHK(root, hipMemsetAsync(root->d_image.p, 0, bytes, root->stream));
  for (int i = 0; i < n; i++) {
    int tiles = (int)(yh_shard_pixels(root, i, n) / 64);
    int e = yhk_unpack((char*)root->d_gather_recv.p + (size_t)i * cap_bytes, i, n, tiles, root->num_tiles_total, root->state.tiles_x, root->state.width,
        root->state.height, root->d_image.p, root->stream);
    if (e) return fail(root, YH_E_DEVICE, "k_unpack launch: %s", hipGetErrorString((hipError_t)e));
  }
  HIPCHK(root, hipMemcpyAsync(rgba, root->d_image.p, bytes, hipMemcpyDeviceToHost, root->stream));
  YH_WAIT(root);
  return YH_OK;
}
