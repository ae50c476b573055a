// context.cpp — the C ABI of include/yhair.h on top of the HIP kernels.
//
// Host responsibilities (all once per scene / per image, never per sample):
//   yh_upload_scene = init_bvh + init_lights (pt.cpp:755-818,1695-1740):
//     reference-identical BVHs, leaf-ordered primitive records, inverse object
//     frames, per-material hair constants (the material-only part of
//     eval_hair_brdf, ext.cpp:131-172), light CDFs, float4 env texels;
//   yh_init_state (pt.cpp:1931-1946): image size, tile list of this shard,
//     per-pixel PCG32 streams (the sequence ids come from ONE serial master
//     generator, so they are produced on the host and uploaded: 16 B/pixel);
//   yh_trace_samples: one k_trace launch on the context's stream, HIP-event
//     timed.
// There is no CPU fallback: without a GPU yh_create returns NULL.
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>  // types only: the library is opened on first use (yh_gather_framebuffer)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <tuple>
#include <string>
#include <memory>
#include <thread>
#include <vector>

#include "../csrc/yh_device.h"
#include "bvh_build.h"
#include "build_id.h"  // YH_BUILD_ID: a hash of the device and host sources, written by the Makefile
#include "yhair.h"

// launchers in csrc/kernels.hip
extern "C" {
int yhk_trace(const yhd_scene*, const yhd_state*, int, yhd_counters*, int shape, int grid_blocks, hipStream_t);
int yhk_trace_exact(const yhd_scene*, const yhd_state*, int nsamples, int lds_bytes, int grid_blocks, hipStream_t);  // csrc/exact.hip
int yhk_trace_exact_occupancy(int lds_bytes, int general);
int yhk_block_threads(int shape);
int yhk_trace_occupancy(int lds_bytes, int general, int shape);
int yhk_trace_sbs(const yhd_scene*, const yhd_state*, int nsamples, int oct_blocks, int quad_items, int oct_entries, int grid_blocks, hipStream_t);
int yhk_trace_sbs_lds_bytes(const yhd_scene* sc);
int yhk_trace_sbs_occupancy(int lds_bytes, int general);
int yhk_trace_lds_bytes(const yhd_scene* sc, int shape);
int yhk_stack_entries(void);
#ifdef YH_LAB_WAVEFRONT  // developer build (make WAVEFRONT=1): the workgroup-staged kernel of csrc/lab/, YHAIR_SHAPE=2
int yhk_wavefront(const yhd_scene*, const yhd_state*, int, const yhd_pool*, int k, int grid_blocks, hipStream_t);
int yhk_wavefront_slots(int k);
int yhk_wavefront_lds_bytes(int stack_entries, int tables_f4, int k);
int yhk_wavefront_occupancy(int lds_bytes, int general, int k);
#endif
int yhk_stream(const yhd_scene*, const yhd_scene* sc_dev, const yhd_state*, int, const yhd_stream*, int grid_blocks, hipStream_t);
int yhk_lane_blob_shape(const yhd_float4* nodes, const yhd_float4* prims, yhd_float4* blob, int kind, int node_base, int num_nodes, int prim_base,
    int num_prims, long long node_off, long long test_off, hipStream_t);
int yhk_stream_block_threads(void);
int yhk_stream_lds_bytes(int tables_f4, int slots_per_wave);
int yhk_stream_occupancy(int lds_bytes, int general);
int yhk_intersect_lanes_occupancy(const yhd_scene* sc, int waves);
int yhk_intersect_lanes(const yhd_scene* sc, const yhd_scene* sc_dev, int n, const float* rays, int* cursor, unsigned int* stack_ovf,
    int ovf_entries, int* object, int* element, float* uv, float* dist, int waves, int grid_blocks, hipStream_t stream);
int yhk_resolve(const yhd_state*, int, int, void*, hipStream_t);
int yhk_pack(const yhd_state*, int, int, void*, hipStream_t);
int yhk_unpack(const void*, int, int, int, int, int, int, int, void*, hipStream_t);
int yhk_hair_brdf(int, const void*, const float*, const float*, const float*, float*, hipStream_t);
int yhk_hair_eval(int, const float*, const float*, const float*, float*, hipStream_t);
int yhk_hair_pdf(int, const float*, const float*, const float*, float*, hipStream_t);
int yhk_hair_sample(int, const float*, const float*, const float*, float*, hipStream_t);
int yhk_intersect(const yhd_scene*, int, const float*, int*, int*, float*, float*, hipStream_t);
int yhk_bvh_build_gpu(int n, const float* boxes, float* nodes8, int* primitives, int* num_nodes, int* depth, hipStream_t);
int yhk_curves_to_lines(int, const float*, const float*, const float*, int, float*, float*, float*, int*, hipStream_t);
int yhk_surface_lobe(int, int, const float*, const float*, const float*, const float*, const float*, float*, hipStream_t);
int yhk_surface_bsdf(int, const void*, const float*, const float*, const float*, const float*, float*, hipStream_t);
int yhk_selftest(int, float, float, uint64_t, uint64_t, int, const float*, double*, unsigned int*, hipStream_t);
}

namespace {

std::string g_create_error = "no error";

struct DevBuf {
  void*  p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;  // owns a hipMalloc pointer
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr, o.bytes = 0; }
  ~DevBuf() { reset(); }
  void reset() {
    if (p) (void)hipFree(p);
    p = nullptr, bytes = 0;
  }
};

const float pif = (float)3.14159265358979323846;

// ---- tiny host vector helpers with the reference's operation order --------
struct F3 {
  float x, y, z;
};
F3    operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
F3    operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
F3    operator-(F3 a) { return {-a.x, -a.y, -a.z}; }
F3    operator*(F3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
F3    cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
float fmin_(float a, float b) { return (a < b) ? a : b; }
float fmax_(float a, float b) { return (a > b) ? a : b; }
F3    ld3(const float* p) { return {p[0], p[1], p[2]}; }
void  st3(float* p, F3 a) { p[0] = a.x, p[1] = a.y, p[2] = a.z; }

// inverse(frame, non_rigid = true) (math.h:2877-2885, 2721-2741)
void inverse_frame(const float* f, bool non_rigid, float* out) {
  F3 x = ld3(f), y = ld3(f + 3), z = ld3(f + 6), o = ld3(f + 9);
  F3 rx, ry, rz;
  if (non_rigid) {
    F3    c0 = cross(y, z), c1 = cross(z, x), c2 = cross(x, y);
    float det = dot(x, cross(y, z));
    float s   = 1 / det;
    rx = F3{c0.x, c1.x, c2.x} * s, ry = F3{c0.y, c1.y, c2.y} * s, rz = F3{c0.z, c1.z, c2.z} * s;
  } else {
    rx = {x.x, y.x, z.x}, ry = {x.y, y.y, z.y}, rz = {x.z, y.z, z.z};
  }
  F3 ro = -(rx * o.x + ry * o.y + rz * o.z);
  st3(out, rx), st3(out + 3, ry), st3(out + 6, rz), st3(out + 9, ro);
}
F3 transform_point(const float* f, F3 b) {
  return ld3(f) * b.x + ld3(f + 3) * b.y + ld3(f + 6) * b.z + ld3(f + 9);
}

// PCG32 (math.h:1396-1442) for init_state and the self-test drivers
struct Rng {
  uint64_t state, inc;
};
uint32_t advance_rng(Rng& rng) {
  uint64_t old        = rng.state;
  rng.state           = old * 6364136223846793005ULL + rng.inc;
  uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot        = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
Rng make_rng(uint64_t seed, uint64_t seq = 1) {
  Rng rng{0, (seq << 1u) | 1u};
  advance_rng(rng);
  rng.state += seed;
  advance_rng(rng);
  return rng;
}
float rand1f(Rng& rng) {
  uint32_t u = (advance_rng(rng) >> 9) | 0x3f800000u;
  float    f;
  memcpy(&f, &u, 4);
  return f - 1.0f;
}
void skip_rng(Rng& rng, uint64_t delta) {  // LCG jump-ahead
  uint64_t cur_mult = 6364136223846793005ULL, cur_plus = rng.inc, acc_mult = 1u, acc_plus = 0u;
  while (delta > 0) {
    if (delta & 1) acc_mult *= cur_mult, acc_plus = acc_plus * cur_mult + cur_plus;
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta /= 2;
  }
  rng.state = acc_mult * rng.state + acc_plus;
}

float sqr(float v) { return v * v; }
template <int N>
float powt(float v) {  // ext.cpp:95-109
  if constexpr (N == 0) return 1;
  else if constexpr (N == 1) return v;
  else {
    float n2 = powt<N / 2>(v);
    return n2 * n2 * powt<(N & 1)>(v);
  }
}

// The material-only part of eval_hair_brdf (ext.cpp:131-172) plus the
// per-lobe constants the kernels use (dev_hair.h). Same libm as the reference
// (this runs on the host), so these values are bit-identical to what the
// reference recomputes at every hit.
void make_material(const yh_material& m, yhd_material& d) {
  memset(&d, 0, sizeof(d));
  memcpy(d.emission, m.emission, 12);
  memcpy(d.color, m.color, 12);
  float dmax    = fmax_(fmax_(m.color[0], m.color[1]), m.color[2]);
  d.diffuse_pdf = dmax ? dmax / dmax : 0.0f;  // pt.cpp:456-471 with one lobe
  d.thin        = m.thin;
  d.specular = m.specular, d.metallic = m.metallic, d.roughness = m.roughness, d.ior = m.ior;
  d.transmission = m.transmission;
  d.opacity      = m.opacity * ((1.0f + 1.0f + 1.0f) / 3);  // mean of the {1,1,1} null texture (pt.cpp:425)
  if (d.opacity > 0.999f) d.opacity = 1;
  d.emission_tex = m.emission_tex - 1, d.color_tex = m.color_tex - 1, d.scattering_tex = m.scattering_tex - 1;
  d.trdepth = m.trdepth;
  d.plain = m.specular == 0 && m.metallic == 0 && m.transmission == 0 && d.opacity == 1 && m.emission_tex == 0 &&
            m.color_tex == 0 && m.scattering_tex == 0;
  for (int c = 0; c < 3; c++) {  // reflectivity_to_eta (math.h:4270-4273)
    float r   = fmin_(fmax_(m.color[c], 0.0f), 0.99f);
    d.meta[c] = (1 + std::sqrt(r)) / (1 - std::sqrt(r));
  }
  d.has_volume = !m.thin && m.transmission != 0;
  for (int c = 0; c < 3; c++) {  // eval_vsdf (pt.cpp:520-524)
    d.vol_density[c] = d.has_volume ? -std::log(fmin_(fmax_(m.color[c], 0.0001f), 1.0f)) / m.trdepth : 0.0f;
    d.vol_scatter[c] = m.scattering[c];
  }
  d.vol_anisotropy = m.scanisotropy;
  F3 sa{0, 0, 0};
  if (m.sigma_a[0] || m.sigma_a[1] || m.sigma_a[2]) {
    sa = ld3(m.sigma_a);
  } else if (m.color[0] || m.color[1] || m.color[2]) {  // ext.cpp:121-125
    float bn  = m.beta_n;
    float den = 5.969f - 0.215f * bn + 2.532f * sqr(bn) - 10.73f * powt<3>(bn) + 5.574f * powt<4>(bn) +
                0.245f * powt<5>(bn);
    F3 q = {std::log(m.color[0]) / den, std::log(m.color[1]) / den, std::log(m.color[2]) / den};
    sa   = {q.x * q.x, q.y * q.y, q.z * q.z};
  } else if (m.eumelanin || m.pheomelanin) {  // ext.cpp:115-119
    F3 e = F3{0.419f, 0.697f, 1.37f}, p = F3{0.187f, 0.4f, 1.05f};
    sa   = F3{m.eumelanin * e.x, m.eumelanin * e.y, m.eumelanin * e.z} +
         F3{m.pheomelanin * p.x, m.pheomelanin * p.y, m.pheomelanin * p.z};
  }
  st3(d.sigma_a, sa);
  d.alpha = m.alpha, d.eta = m.eta;
  float bm = m.beta_m, bn = m.beta_n;
  d.v[0] = sqr(0.726f * bm + 0.812f * sqr(bm) + 3.7f * powt<20>(bm));
  d.v[1] = 0.25f * d.v[0];
  d.v[2] = 4 * d.v[0];
  d.v[3] = d.v[2];
  d.s    = 0.626657069f * (0.265f * bn + 1.194f * sqr(bn) + 5.372f * powt<22>(bn));
  d.sin_2k_alpha[0] = std::sin(pif / 180 * d.alpha);
  d.cos_2k_alpha[0] = std::sqrt(fmax_(0.0f, 1 - sqr(d.sin_2k_alpha[0])));
  for (int i = 1; i < 3; i++) {
    d.sin_2k_alpha[i] = 2 * d.cos_2k_alpha[i - 1] * d.sin_2k_alpha[i - 1];
    d.cos_2k_alpha[i] = sqr(d.cos_2k_alpha[i - 1]) - sqr(d.sin_2k_alpha[i - 1]);
  }
  for (int p = 0; p < 4; p++) {
    d.inv_v[p]        = 1 / d.v[p];
    d.log_inv_2v[p]   = std::log(1 / (2 * d.v[p]));
    d.exp_m2_inv_v[p] = std::exp(-2 / d.v[p]);
    d.mp_den[p]       = ::sinh((double)(1 / d.v[p])) * 2 * d.v[p];
  }
  float cb   = 1 / (1 + std::exp(-pif / d.s));
  float ca   = 1 / (1 + std::exp(-(-pif) / d.s));
  d.tl_cdf_a = ca;
  d.tl_norm  = cb - ca;
}

}  // namespace

// Splits [0, n) over a few host threads (upload-time array fills; not a hot path).
template <typename F>
void parallel_for(int n, F&& fn) {
  int nt = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (n < 65536 || nt == 1) {
    for (int i = 0; i < n; i++) fn(i);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; t++)
    pool.emplace_back([=, &fn] {
      int lo = (int)((int64_t)n * t / nt), hi = (int)((int64_t)n * (t + 1) / nt);
      for (int i = lo; i < hi; i++) fn(i);
    });
  for (auto& th : pool) th.join();
}

constexpr int YH_SHAPES = 9;  // launch shapes: 0, 1 k_trace (4-wide nodes) | 2 k_trace over 8-wide nodes | 3 k_stream | 4 k_trace with octets | 5 quads and octets side by side | 6 k_trace with sixteen lanes per path | 7 octets with leaf pairs | 8 sixteen lanes with leaf groups
// The 8- and 16-wide collapses of a scene's trees take a tenth of a second of host time for a million-segment hair
// model; the kernels that need them are chosen after the first launches. yh_upload_scene starts them in the background,
// ensure_wide_nodes (host/context.cpp, below) waits for them — so the first launch of a wide kernel does not pay for them.
struct WideBuild {
  std::thread                               th;
  std::vector<std::vector<yhh::WideNode8>>  w8;
  std::vector<std::vector<yhh::WideNode16>> w16;
  std::vector<int>                          d8, d16;
};
struct yh_context {
  int         device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t  ev0 = nullptr, ev1 = nullptr;
  int         hy_quad_items = 0, hy_oct_entries = 0;  // layout of the work list for shape 5: [quad items][octet entries]
  std::vector<int> hy_oct_items;                       // ... and the items that run as octets
  int         num_cus = 0;
  std::string device_name;  // gcnArchName / marketing name / CU count: part of the key of the trial record on disk
  std::string error = "no error";
  // scene
  bool      have_scene = false;
  yhd_scene scene{};
  DevBuf    d_nodes, d_nodes8, d_nodes16, d_prims, d_vpos, d_elems, d_objects, d_materials, d_scene_nodes,
      d_scene_prims, d_light_cdf, d_env_texels, d_light_table, d_env_tab;
  int       stack_need = 0, stack_need8 = 0, stack_need16 = 0;
  // The 8- and 16-wide node arrays (launch shapes 4, 6, 7) are built and uploaded at their first use (ensure_wide_nodes):
  // an image that never runs those kernels pays neither the collapses nor the memory. Until then the host keeps the
  // shapes' binary trees and the object records.
  bool                     wide_built = false;
  std::unique_ptr<struct WideBuild> wide_job;  // the collapses of host_trees, started in the background by yh_upload_scene
  std::vector<yhh::Tree>   host_trees;    // per shape (emptied once the wide arrays exist)
  std::vector<yhd_object>  host_objects;  // as uploaded; wbox_min[3] / wbox_max[3] = the wide arrays' bases once built
  std::vector<int>         object_shape;  // shape index of every object
  // the one-lane kernels' copy of the trees (yhd_scene::lane_blob): laid out at upload, filled on the device at the first
  // launch of k_stream / k_intersect_lanes (ensure_lane_blob)
  struct LaneShape { int kind, node_base, num_nodes, prim_base, num_prims; long long node_off, test_off; };  // offsets in 32-byte units
  std::vector<LaneShape>   lane_shapes;
  long long                lane_units = 0;
  DevBuf                   d_lane_blob;
  // state
  bool             have_state = false;
  yhd_state        state{};
  yh_trace_params  params{};
  DevBuf           d_textures, d_tex_texels, d_vtex;
  DevBuf           d_rng_state, d_rng_inc, d_accum, d_tiles, d_image, d_counters, d_tile_cursor, d_tile_cost;
  std::vector<int> owned;      // owned tile ids, increasing
  std::vector<unsigned int>  item_cost;  // per work item (tile * 4 + quadrant): last measured cost (scheduling hint, kept across init_state)
  int              rank = 0, world = 1;
  int              num_tiles_total = 0;
  float            last_ms = 0;
  int              last_launches = 0;
  unsigned         launches_of_state = 0;  // synchronous launches since yh_init_state (re-planning schedule)
  int              launch_shape = 0;  // decided from launches of at least 16 spp (shorter ones have flat, noisy item costs)
  int              last_shape = -1;   // the kernel the most recent launch ran (yh_launch_shape)
  bool             async_pending = false;  // an asynchronous launch whose time yh_synchronize has still to read
  bool             last_counted = false;  // ... and whether it was the instrumented build (its time ranks nothing)
  // single-process multi-GPU gather (yh_gather_framebuffer): this context's packed tiles; on the root also the
  // receive buffer and the communicators of the device set they were made for
  DevBuf                  d_gather_send, d_gather_recv;
  std::vector<ncclComm_t> comms;
  std::vector<int>        comm_devices;
  // kernel selection by measurement (pick_launch_shape): ms per sample of a planned launch with each kernel
  // (0 = not measured yet), whether item costs exist (the first launch of a scene runs unplanned and is not a
  // measurement), and whether the scene is dense (more expensive items than resident waves; from k_trace's costs)
  double           shape_ms[YH_SHAPES] = {};   // (indexed by launch shape, yhd_state::launch_shape)
  int              shape_trials[YH_SHAPES] = {};  // trial launches behind each shape_ms (the minimum over them counts)
  uint64_t         scene_key = 0;                   // fingerprint of the uploaded scene (key of the process-wide trial record)
  bool             trials_from_disk = false;        // the record was read from the on-disk cache: complete, no trial runs
  bool             have_costs = false;
  bool             costs_settled = false;   // the item costs come from a launch of at least YH_TRIAL_SPP samples (not from the 1-spp probe)
  bool             planned_settled = false; // ... and the most recent launch was planned from such costs (only then does its time rank a kernel)
  int              dense = -1;
  int              chain16 = -1; // 1: ... and four times as many: the sixteen-lane form (shape 6) is a candidate too
  int              chain = -1;   // 1: so few expensive items that even twice as many waves would all be resident: the launch is bound by the
                                 // chain of steps of ONE path, and the octet kernel (half the paths per wave, shape 4) is a candidate
  // path pool of the wavefront integrator (csrc/wavefront.hip), allocated at its first launch
  DevBuf           d_pool_ray_o, d_pool_ray_d, d_pool_weight, d_pool_radiance, d_pool_hit, d_pool_medium;
  // path pool of the streaming integrator (csrc/stream.hip): per-wave slots, allocated at its first launch
  DevBuf           d_st_slots, d_st_medium, d_st_ovf, d_st_prof, d_scene_copy;
  size_t           st_slots = 0, st_medium_slots = 0, st_ovf_words = 0;
  yhd_stream       stream_pool{};
  size_t           pool_slots = 0, pool_medium_slots = 0;  // capacity of the per-slot arrays / of the medium array (general scenes only)
  yhd_pool         pool{};
};

namespace {

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
#define HIPCHK(ctx, call)                                                                               \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return fail(ctx, YH_E_DEVICE, "%s: %s", #call, hipGetErrorString(e_));       \
  } while (0)

int upload(yh_context* ctx, DevBuf& buf, const void* src, size_t bytes) {
  buf.reset();
  size_t alloc = std::max<size_t>(bytes, 16);
  HIPCHK(ctx, hipMalloc(&buf.p, alloc));
  buf.bytes = alloc;
  if (bytes) HIPCHK(ctx, hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice));
  return YH_OK;
}
int alloc_zero(yh_context* ctx, DevBuf& buf, size_t bytes) {
  buf.reset();
  size_t alloc = std::max<size_t>(bytes, 16);
  HIPCHK(ctx, hipMalloc(&buf.p, alloc));
  buf.bytes = alloc;
  // The context's stream is non-blocking: a memset on the null stream would not be
  // ordered against kernels launched on it. Clear on that stream and wait, so the
  // buffer is zero for whoever touches it next (stream kernel or blocking copy).
  HIPCHK(ctx, hipMemsetAsync(buf.p, 0, alloc, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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

int tiles_of(int n) { return (n + YH_TILE - 1) / YH_TILE; }

// Work items (yh_device.h: yhd_state::tiles) for the owned tiles with their
// current split modes, most expensive first.
void build_work_items(const yh_context* ctx, std::vector<int>& items);
int  choose_launch_shape(const yh_context* ctx);
void record_launch(yh_context* ctx, int nsamples, bool fresh_costs);
int  pick_launch_shape(const yh_context* ctx, int nsamples);
bool trial_pending(const yh_context* ctx);
bool trials_off();

}  // namespace

namespace {
// Dense or sparse? When every pixel is expensive the quad kernel is latency-bound and more waves per SIMD pay
// (k_trace 256 x 5: C2 +19 %, C4 +10 % over 512 x 4); when a few expensive pixels bound the launch (C1: the hair covers
// 11 % of the frame and barely fills the resident waves) they cost 11 %. Measure: the number of
// max-cost work items the last launch was worth (sum of item costs over the largest) against the
// resident wave slots — measured 3 066 (C1), 4 777 (C4), 8 989 (C2), 24 970 (C3) against 4 096: with
// fewer expensive items than slots every wave that can run already does. This only picks the CANDIDATES; which
// kernel runs is measured (pick_launch_shape). YHAIR_SHAPE=0..3 overrides.
// Is the launch worth more max-cost work items than there are resident waves? (item costs of a k_trace launch)
bool dense_by_costs(const yh_context* ctx, bool* known, bool* chain_bound = nullptr, bool* chain16 = nullptr) {
  uint64_t sum = 0, mx = 0;
  for (int t : ctx->owned)
    for (int p = 0; p < 4; p++) {
      uint64_t c = ctx->item_cost[(size_t)t * 4 + p];
      sum += c, mx = std::max(mx, c);
    }
  *known = mx != 0;
  if (mx == 0) return false;
  // YHAIR_DEVICE_SHARE=k: k processes render on this device at once (bench.py with more ranks than devices): a k-th of the waves is ours
  static const double share = std::max(1, getenv("YHAIR_DEVICE_SHARE") ? atoi(getenv("YHAIR_DEVICE_SHARE")) : 1);
  int    lds      = yhk_trace_lds_bytes(&ctx->scene, 0);
  double resident = (double)ctx->num_cus * std::max(1, yhk_trace_occupancy(lds, ctx->scene.general_materials, 0)) * (yhk_block_threads(0) / 64) / share;
  if (getenv("YHAIR_TIMING")) fprintf(stderr, "[yhair] launch shape: worth %.0f items, resident waves %.0f\n", (double)sum / (double)mx, resident);
  if (chain_bound) {  // the octet kernel needs two waves per expensive item: all of them resident at once, with room to spare
    const int    lds4 = yhk_trace_lds_bytes(&ctx->scene, 4);
    const double res4 = (double)ctx->num_cus * std::max(1, yhk_trace_occupancy(lds4, ctx->scene.general_materials, 4)) * (yhk_block_threads(4) / 64) / share;
    // (candidacy only — the trials decide: generous bounds cost a wasted trial, tight ones a missed kernel; `textured`, whose
    // item costs are very uneven, is worth 1 500 items and still renders 1.45 x faster with sixteen lanes per path)
    *chain_bound      = 2.0 * (double)sum / (double)mx <= 1.1 * res4;  // (C1 at 720^2 is worth 2 400-3 100 items: not one; half of it 1 400-1 700: one)
    if (chain16) *chain16 = 4.0 * (double)sum / (double)mx <= 2.5 * res4;
  }
  return (double)sum / (double)mx >= resident;
}
int choose_launch_shape(const yh_context* ctx) {
  if (const char* env = getenv("YHAIR_SHAPE")) return std::max(0, std::min(YH_SHAPES - 1, atoi(env)));
  bool known = false;
  return dense_by_costs(ctx, &known) ? 1 : 0;
}
// Kernel selection by MEASUREMENT (every kernel renders the same bits, so trying one costs time only). k_trace at
// 512 x 4 suits launches bound by a few expensive pixels (C1), k_trace at 256 x 5 and the one-lane-per-path k_stream
// suit dense scenes, and which of those two wins depends on how many expensive pixels there are per wave
// (straight-hair 720^2: a tie; curly-hair 1280^2: k_stream +39 %; hair-curls: k_trace 2.3x). Every candidate is timed
// once per image on a SHORT planned launch (YH_TRIAL_SPP samples: yh_trace_samples cuts them off the front of a long
// request, so all samples count and a trial of the wrong kernel costs milliseconds — a whole 512-spp launch of it cost
// hair-curls 14 % of an 8-launch render), then the fastest per sample stays. Only launches of that length class rank
// kernels: a short launch costs more per sample than a long one (C1, 512 x 4: 0.25 against 0.23 ms), so a long launch
// of the running kernel must not be compared with the trials of the others; and only launches planned from the item
// costs of a launch of that length or more: the hand-out order planned from the 1-spp probe costs 15 % of a launch
// (C1: 0.269 against 0.234 ms per sample), so on a new image a first short launch settles the costs and the trials
// follow it. Sparse scenes never try k_stream: it costs them a fixed 20 ms per launch for the cheap pixels.
constexpr int YH_TRIAL_SPP = 32;  // shorter launches have flat, noisy costs: they neither rank kernels nor try new ones
// One 32-sample trial is a noisy measurement (± 5 % launch to launch): when the runner-up is within YH_TRIAL_TIE of the
// best, both are tried a second time and the minimum of a kernel's trials counts, so that two ranks rendering halves of
// one image, or two renders of one image, do not settle on different kernels by chance.
constexpr double YH_TRIAL_TIE  = 1.15;
constexpr double YH_FINAL_TIE  = 1.05;  // after the trials: candidates this close to the fastest count as tied (pick_launch_shape)
constexpr int    YH_TRIALS_MAX = 2;
// The trial results of an image are kept per process under (scene fingerprint, image size, shard, bounces): a new
// context on the same scene and image (a re-render, the next frame of a caller that re-creates its context) starts
// from them instead of re-deciding.
struct TrialKey {
  uint64_t scene;
  int      w, h, rank, world, bounces;
  bool operator<(const TrialKey& o) const {
    return std::tie(scene, w, h, rank, world, bounces) < std::tie(o.scene, o.w, o.h, o.rank, o.world, o.bounces);
  }
};
struct TrialRecord {
  double ms[YH_SHAPES];
  int    trials[YH_SHAPES], dense, chain, chain16;
};
std::mutex                      g_trials_mutex;
std::map<TrialKey, TrialRecord> g_trials;
TrialKey trial_key(const yh_context* ctx) {
  return TrialKey{ctx->scene_key, ctx->state.width, ctx->state.height, ctx->rank, ctx->world, ctx->state.bounces};
}
// The same record ON DISK (round 4), so that the kernel an image runs does not depend on a handful of 32-sample launches
// re-decided by every process (every rank of every run): ~/.cache/yhair/trials_v1.txt (YHAIR_CACHE_DIR, XDG_CACHE_HOME),
// one line per record, keyed by device, the build's fingerprint (host/build_id.h: a hash of the device and host sources),
// YHAIR_DEVICE_SHARE and the TrialKey; appended with one O_APPEND write (atomic between the ranks of a run), the last line
// of a key counts. Only COMPLETE records are written (no candidate still wants a trial) and a loaded one is complete by
// construction, so a process that finds its image here runs no trial at all. YHAIR_NO_DISK_CACHE (or YHAIR_NO_TRIAL_CACHE,
// which also forgets the per-process record) switches it off.
std::string disk_cache_path() {
  if (getenv("YHAIR_NO_DISK_CACHE") || getenv("YHAIR_NO_TRIAL_CACHE")) return "";
  std::string dir;
  if (const char* e = getenv("YHAIR_CACHE_DIR")) dir = e;
  else if (const char* x = getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/yhair";
  else if (const char* h = getenv("HOME")) dir = std::string(h) + "/.cache/yhair";
  else return "";
  return dir + "/trials_v1.txt";
}
std::string disk_key(const yh_context* ctx) {
  const char* share = getenv("YHAIR_DEVICE_SHARE");
  char buf[256];
  snprintf(buf, sizeof(buf), "%s|%s|%s|%016llx|%d|%d|%d|%d|%d", ctx->device_name.c_str(), YH_BUILD_ID, share ? share : "1",
      (unsigned long long)ctx->scene_key, ctx->state.width, ctx->state.height, ctx->rank, ctx->world, ctx->state.bounces);
  return buf;
}
void mkdirs(const std::string& file) {
  for (size_t i = 1; i < file.size(); i++)
    if (file[i] == '/') (void)mkdir(file.substr(0, i).c_str(), 0755);
}
void disk_store(const yh_context* ctx, const TrialRecord& r) {
  const std::string path = disk_cache_path();
  if (path.empty()) return;
  mkdirs(path);
  std::string line = disk_key(ctx) + " =";
  char        buf[64];
  for (int k = 0; k < YH_SHAPES; k++) {
    snprintf(buf, sizeof(buf), " %.9g:%d", std::isinf(r.ms[k]) ? -1.0 : r.ms[k], r.trials[k]);
    line += buf;
  }
  snprintf(buf, sizeof(buf), " ; %d %d %d\n", r.dense, r.chain, r.chain16);
  line += buf;
  int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
  if (fd < 0) return;
  (void)!write(fd, line.data(), line.size());
  close(fd);
}
bool disk_load(const yh_context* ctx, TrialRecord& r) {
  const std::string path = disk_cache_path();
  if (path.empty()) return false;
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  const std::string key = disk_key(ctx) + " =";
  bool  found = false;
  char  line[2048];
  while (fgets(line, sizeof(line), f)) {
    if (strncmp(line, key.c_str(), key.size()) != 0) continue;
    TrialRecord t{};
    const char* p  = line + key.size();
    bool        ok = true;
    for (int k = 0; k < YH_SHAPES && ok; k++) {
      int n = 0;
      ok    = sscanf(p, " %lf:%d%n", &t.ms[k], &t.trials[k], &n) == 2;
      p += n;
      if (ok && t.ms[k] < 0) t.ms[k] = std::numeric_limits<double>::infinity();  // a candidate that cannot run on this device
    }
    if (ok && sscanf(p, " ; %d %d %d", &t.dense, &t.chain, &t.chain16) == 3) r = t, found = true;  // (the last line of a key counts)
  }
  fclose(f);
  return found;
}
void trials_store(const yh_context* ctx) {
  TrialRecord r;
  for (int k = 0; k < YH_SHAPES; k++) r.ms[k] = ctx->shape_ms[k], r.trials[k] = ctx->shape_trials[k];
  r.dense = ctx->dense, r.chain = ctx->chain, r.chain16 = ctx->chain16;
  {
    std::lock_guard<std::mutex> lock(g_trials_mutex);
    g_trials[trial_key(ctx)] = r;
  }
  // on disk only what a later process may rely on: the choice was made by the trials (no forced shape, no heuristic-only
  // mode), dense / sparse is known, and nothing is left to try on this image
  if (!trials_off() && ctx->dense >= 0 && ctx->costs_settled && !trial_pending(ctx)) disk_store(ctx, r);
}
void trials_load(yh_context* ctx) {
  ctx->trials_from_disk = false;
  if (getenv("YHAIR_NO_TRIAL_CACHE")) return;  // developer switch
  TrialRecord r{};
  bool        have = false;
  {
    std::lock_guard<std::mutex> lock(g_trials_mutex);
    auto it = g_trials.find(trial_key(ctx));
    if (it != g_trials.end()) r = it->second, have = true;
  }
  if (!have && !trials_off() && disk_load(ctx, r)) {
    have = true;
    ctx->trials_from_disk = true;
    if (getenv("YHAIR_TIMING")) fprintf(stderr, "[yhair] kernel trials of this image: read from %s\n", disk_cache_path().c_str());
  }
  if (!have) return;
  for (int k = 0; k < YH_SHAPES; k++) ctx->shape_ms[k] = r.ms[k], ctx->shape_trials[k] = r.trials[k];
  ctx->dense = r.dense, ctx->chain = r.chain, ctx->chain16 = r.chain16;
}
bool trials_off() {
  static const bool off = getenv("YHAIR_NO_TRIALS") != nullptr;  // developer switch: the cost heuristic only
  return off || getenv("YHAIR_SHAPE") != nullptr;
}
// k_trace 512 x 4 always; the dense quad shape unless the image is chain-bound; k_stream on dense images; the side-by-side
// launch on sparse ones; on chain-bound
// ones (a shard of a sparse image on one of several GPUs, a small image) the octet kernel and, when even four waves per
// expensive item are all resident, the sixteen-lane one. (Shape 2 is never tried: profiles/r03/.)
int candidates(const yh_context* ctx, int cand[6]) {
  int n = 0;
  cand[n++] = 0;
  if (ctx->chain > 0 && ctx->dense <= 0) {  // chain-bound: more lanes per path for every item (the dense quad shape and the side-by-side launch are not tried there)
    cand[n++] = 4, cand[n++] = 7;  // octets, without and with leaf pairs (which of the two wins depends on the share of leaf steps)
    if (ctx->chain16 > 0) cand[n++] = 6, cand[n++] = 8;  // (likewise without and with leaf groups)
    return n;
  }
  if (ctx->dense == 0) cand[n++] = 5;  // sparse, not chain-bound: the few items that top every launch as octets beside the quads (side by side in one launch)
  cand[n++] = 1;
  if (ctx->dense > 0) cand[n++] = 3;
  return n;
}
// After a synchronous launch: its time if it was a trial-length one, and dense / sparse from fresh item costs of a
// k_trace launch.
void record_launch(yh_context* ctx, int nsamples, bool fresh_costs) {
  const int last = ctx->last_shape;
  bool trial = false;
  static const bool prof_build = getenv("YHAIR_ST_PROF") && atoi(getenv("YHAIR_ST_PROF")) != 0;  // the instrumented k_stream: its times rank nothing
  if (!prof_build && nsamples >= YH_TRIAL_SPP && nsamples < 2 * YH_TRIAL_SPP && ctx->planned_settled && !ctx->last_counted && !ctx->params.hair_exact && last >= 0 && last < YH_SHAPES && ctx->last_ms > 0) {
    const double ms = (double)ctx->last_ms / nsamples;
    ctx->shape_ms[last] = ctx->shape_trials[last] > 0 ? std::min(ctx->shape_ms[last], ms) : ms;
    ctx->shape_trials[last]++;
    trial = true;
  }
  if (fresh_costs && nsamples >= YH_TRIAL_SPP) ctx->costs_settled = true;
  // dense / sparse from the item costs of a k_trace launch long enough to mean something: a trial-length launch, or —
  // while nothing is known yet — one of a few samples (the 1-spp probe's costs are too flat to decide on)
  if (fresh_costs && (last == 0 || last == 1) && (nsamples >= YH_TRIAL_SPP || (ctx->dense < 0 && nsamples >= 4))) {
    bool known = false, chain = false, chain16 = false, d = dense_by_costs(ctx, &known, &chain, &chain16);
    if (known) ctx->dense = d ? 1 : 0, ctx->chain = (!d && chain) ? 1 : 0, ctx->chain16 = (!d && chain16) ? 1 : 0;
  }
  ctx->have_costs = true;
  if (trial) trials_store(ctx);
}
// Does candidate c want a (further) trial? Untimed: yes. Timed once: when it is one of at least two candidates within
// YH_TRIAL_TIE of the best (a tie at the noise of one trial).
bool wants_trial(const yh_context* ctx, const int* cand, int n, int c) {
  if (ctx->shape_ms[c] == 0) return true;
  if (ctx->shape_trials[c] >= YH_TRIALS_MAX) return false;
  double best = 0;
  for (int k = 0; k < n; k++) {
    if (ctx->shape_ms[cand[k]] == 0) return false;  // (first trials first)
    if (best == 0 || ctx->shape_ms[cand[k]] < best) best = ctx->shape_ms[cand[k]];
  }
  int close = 0;
  for (int k = 0; k < n; k++) close += ctx->shape_ms[cand[k]] <= YH_TRIAL_TIE * best;
  return close >= 2 && ctx->shape_ms[c] <= YH_TRIAL_TIE * best;
}
// Is a candidate kernel still untimed on this image (so that a long request should start with a short trial)?
bool trial_pending(const yh_context* ctx) {
  if (!ctx->have_state || !ctx->have_costs || ctx->state.shader != YH_SHADER_PATH || trials_off() || ctx->params.hair_exact) return false;
  int cand[6], n = candidates(ctx, cand);
  if (ctx->trials_from_disk) {  // a record from the disk cache is complete: no settling launch, no trial — unless the candidates have changed
    bool complete = true;
    for (int k = 0; k < n; k++) complete = complete && ctx->shape_ms[cand[k]] != 0;
    if (complete) return false;
  }
  if (!ctx->costs_settled) return true;  // (the first short launch settles the item costs; the trials follow it)
  for (int k = 0; k < n; k++)
    if (wants_trial(ctx, cand, n, cand[k])) return true;
  return false;
}
// The kernel for a launch of `nsamples`.
int pick_launch_shape(const yh_context* ctx, int nsamples) {
  if (ctx->params.hair_exact) return 0;  // the exact arithmetic exists as the 512 x 4 quad kernel only (csrc/exact.hip)
  if (const char* env = getenv("YHAIR_SHAPE")) return std::max(0, std::min(YH_SHAPES - 1, atoi(env)));
  if (!ctx->have_costs) return ctx->launch_shape;  // the first launch of an image: unplanned, not a measurement
  const int by_costs = ctx->dense > 0 ? 1 : 0;
  if (trials_off()) return by_costs;
  int cand[6], n = candidates(ctx, cand), best = -1;
  const bool trial_length = ctx->costs_settled && nsamples >= YH_TRIAL_SPP && nsamples < 2 * YH_TRIAL_SPP && !(ctx->trials_from_disk && !trial_pending(ctx));
  for (int k = 0; k < n; k++) {
    const int c = cand[k];
    if (trial_length && wants_trial(ctx, cand, n, c)) return c;  // a trial
    if (ctx->shape_ms[c] == 0) continue;
    if (best < 0 || ctx->shape_ms[c] < ctx->shape_ms[best]) best = c;
  }
  if (best < 0) return by_costs;
  // A tie is decided by a FIXED order, not by the noise of the last 32-sample launch: among the candidates within
  // YH_FINAL_TIE of the fastest the first of k_stream, the dense quad shape, the side-by-side launch, the wide forms
  // (leaf groups before plain), the plain quad kernel — so that two renders (two ranks, two boxes) of one image run the same kernel.
  static const int order[YH_SHAPES] = {3, 1, 5, 8, 7, 6, 4, 0, 2};
  for (int o = 0; o < YH_SHAPES; o++)
    for (int k = 0; k < n; k++)
      if (cand[k] == order[o] && ctx->shape_ms[cand[k]] != 0 && ctx->shape_ms[cand[k]] <= YH_FINAL_TIE * ctx->shape_ms[best]) return cand[k];
  return best;
}
void build_work_items(const yh_context* ctx, std::vector<int>& items) {
  // Expensive items first, in decreasing cost (they bound the launch); the cheap
  // majority (background quadrants, within 8x of the median) follows unsorted:
  // its order does not matter and sorting it would cost more than it saves.
  std::vector<uint64_t> keys;
  keys.reserve(ctx->owned.size() * 4);
  for (int t : ctx->owned)
    for (int p = 0; p < 4; p++) {
      unsigned item = (unsigned)(t * 4 + p);
      keys.push_back(((uint64_t)(0xFFFFFFFFu - ctx->item_cost[item]) << 32) | item);
    }
  if (!keys.empty()) {
    auto mid = keys.begin() + keys.size() / 2;
    std::nth_element(keys.begin(), mid, keys.end());
    uint64_t median_cost = 0xFFFFFFFFu - (uint32_t)(*mid >> 32);
    uint64_t cut_cost    = std::min<uint64_t>(0xFFFFFFFFu, median_cost * 8 + 1);
    uint64_t cut_key     = (uint64_t)(0xFFFFFFFFu - (uint32_t)cut_cost) << 32;  // keys below it cost more than cut_cost
    auto heavy_end = std::partition(keys.begin(), keys.end(), [&](uint64_t k) { return k < cut_key; });
    std::sort(keys.begin(), heavy_end);
  }
  items.resize(keys.size());
  for (size_t i = 0; i < keys.size(); i++) items[i] = (int)(keys[i] & 0xFFFFFFFFu);
}
}  // namespace

extern "C" {

static void deal_items_for_stream(yh_context* ctx, std::vector<int>& items);
// The octet kernel (launch shape 4: eight lanes per path) takes HALF a quadrant per wave: entry = item << 1 | half, the two
// halves of an item next to each other in the cost-sorted order.
static void split_items_for_octets(std::vector<int>& items) {
  std::vector<int> out;
  out.reserve(items.size() * 2);
  for (int it : items) out.push_back(it << 1), out.push_back((it << 1) | 1);
  items.swap(out);
}
// ... and the sixteen-lane form (shape 6) a QUARTER: entry = item << 2 | row of the 4x4 block.
static void split_items_for_hex(std::vector<int>& items) {
  std::vector<int> out;
  out.reserve(items.size() * 4);
  for (int it : items)
    for (int k = 0; k < 4; k++) out.push_back((it << 2) | k);
  items.swap(out);
}
static void split_items_side_by_side(yh_context* ctx, std::vector<int>& items);
static int  ensure_wide_nodes(yh_context* ctx);
static int  ensure_lane_blob(yh_context* ctx);
static void wide_build_join(yh_context* ctx);
static void wide_build_start(yh_context* ctx);
static void lay_out_first_round(const yh_context* ctx, std::vector<int>& items, int shape);
static void lay_out_range(const yh_context* ctx, int* items, size_t n, int wpb, int G, int block_offset = 0);
static int  trace_impl(yh_context* ctx, int nsamples, bool counted, bool sync);

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

static void destroy_communicators(yh_context* ctx);
void yh_destroy(yh_context* ctx) {
  if (!ctx) return;
  if (ctx->wide_job && ctx->wide_job->th.joinable()) ctx->wide_job->th.join();
  destroy_communicators(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* yh_last_error(const yh_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

static int build_bvh_device(yh_context* ctx, const std::vector<yhh::Box>& boxes, yhh::Tree& tree);

int yh_upload_scene(yh_context* ctx, const yh_scene_desc* sd) {
  if (!ctx) return YH_E_INVALID;
  if (!sd) return fail(ctx, YH_E_INVALID, "scene is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (sd->num_objects <= 0) return fail(ctx, YH_E_INVALID, "scene has no objects");
  if (sd->num_environments > YH_MAX_ENVS) return fail(ctx, YH_E_INVALID, "more than %d environments", YH_MAX_ENVS);
  // YHAIR_TIMING=1: stage times of the upload on stderr
  const bool timing = getenv("YHAIR_TIMING") && atoi(getenv("YHAIR_TIMING")) != 0;
  auto       t_last = std::chrono::steady_clock::now();
  auto       lap    = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[yhair] upload: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // ---- per-shape BVHs and flattened arrays --------------------------------
  struct ShapeInfo {
    int kind, node_base, prim_base, vert_base, elem_base, has_normals, depth;
    int node8_base, depth8;  // the same tree collapsed three levels at a time (yhd_scene::nodes8)
    int node16_base, depth16;  // ... and four (yhd_scene::nodes16)
    yhh::Box root;
    int num_nodes, num_prims;
  };
  std::vector<ShapeInfo>  info(sd->num_shapes);
  std::vector<yhd_float4> nodes, prims, vpos;
  wide_build_join(ctx), ctx->wide_job.reset();  // (a previous scene's collapses may still be running on the trees replaced below)
  ctx->wide_built = false;
  ctx->host_trees.assign((size_t)sd->num_shapes, yhh::Tree{});
  ctx->d_nodes8.reset(), ctx->d_nodes16.reset();
  std::vector<float>      vtex;  // 2 per vertex, zeros for shapes without texture coordinates
  std::vector<yhd_int4>   elems;
  int                     best_lines = -1, best_shape = -1;
  {  // one allocation per array: growing them shape by shape would re-copy the hair every time
    size_t np = 0, nv = 0, ne = 0;
    for (int si = 0; si < sd->num_shapes; si++) {
      auto& s = sd->shapes[si];
      bool  lines = s.num_lines > 0;
      size_t nel = (size_t)std::max(0, lines ? s.num_lines : s.num_triangles);
      np += nel * (lines ? 4 : 6), nv += (size_t)std::max(0, s.num_vertices), ne += nel;
    }
    prims.reserve(np), vpos.reserve(nv), vtex.reserve(2 * nv), elems.reserve(ne), nodes.reserve(ne * 6);
  }
  for (int si = 0; si < sd->num_shapes; si++) {
    auto& s = sd->shapes[si];
    if (s.num_vertices <= 0 || !s.positions) return fail(ctx, YH_E_INVALID, "shape %d has no vertices", si);
    bool lines = s.num_lines > 0;
    if (!lines && s.num_triangles <= 0) return fail(ctx, YH_E_INVALID, "shape %d has no lines or triangles", si);
    int nel = lines ? s.num_lines : s.num_triangles;
    // a leaf reference packs its first record into 27 bits (host/bvh_build.cpp: count << 27 | start)
    if (nel >= (1 << 27)) return fail(ctx, YH_E_INVALID, "shape %d has %d elements (limit %d)", si, nel, (1 << 27) - 1);
    const int* idx = lines ? s.lines : s.triangles;
    for (int k = 0; k < nel * (lines ? 2 : 3); k++)
      if (idx[k] < 0 || idx[k] >= s.num_vertices) return fail(ctx, YH_E_INVALID, "shape %d: vertex index out of range", si);
    auto& I       = info[si];
    I.kind        = lines ? YH_KIND_LINES : YH_KIND_TRIANGLES;
    I.node_base   = (int)nodes.size() / 8;
    I.prim_base   = (int)prims.size();
    I.vert_base   = (int)vpos.size();
    I.elem_base   = (int)elems.size();
    I.has_normals = s.normals != nullptr;
    auto pos = [&](int v) { return ld3(s.positions + 3 * (size_t)v); };
    auto rad = [&](int v) { return s.radius ? s.radius[v] : 0.001f; };  // add_radius, sceneio.cpp:390
    std::vector<yhh::Box> boxes(nel);
    parallel_for(nel, [&](int e) {
      if (lines) {  // line_bounds (math.h:3037-3040)
        int a = idx[2 * e], b = idx[2 * e + 1];
        F3  p0 = pos(a), p1 = pos(b);
        float r0 = rad(a), r1 = rad(b);
        float lo0[3] = {p0.x - r0, p0.y - r0, p0.z - r0}, lo1[3] = {p1.x - r1, p1.y - r1, p1.z - r1};
        float hi0[3] = {p0.x + r0, p0.y + r0, p0.z + r0}, hi1[3] = {p1.x + r1, p1.y + r1, p1.z + r1};
        for (int k = 0; k < 3; k++) boxes[e].min[k] = fmin_(lo0[k], lo1[k]), boxes[e].max[k] = fmax_(hi0[k], hi1[k]);
      } else {  // triangle_bounds (math.h:3041-3044)
        const float* p0 = s.positions + 3 * (size_t)idx[3 * e];
        const float* p1 = s.positions + 3 * (size_t)idx[3 * e + 1];
        const float* p2 = s.positions + 3 * (size_t)idx[3 * e + 2];
        for (int k = 0; k < 3; k++) {
          boxes[e].min[k] = fmin_(p0[k], fmin_(p1[k], p2[k]));
          boxes[e].max[k] = fmax_(p0[k], fmax_(p1[k], p2[k]));
        }
      }
    });
    lap("primitive bounds");
    yhh::Tree tree;
    // big shapes: the same tree, built on the device (YHAIR_BVH=host forces the host builder)
    static const bool host_only = getenv("YHAIR_BVH") && !strcmp(getenv("YHAIR_BVH"), "host");
    if (nel >= 32768 && !host_only) {
      int rc = build_bvh_device(ctx, boxes, tree);
      if (rc) return rc;
    } else {
      yhh::build_bvh(tree, boxes);
    }
    lap("build_bvh (reference tree)");
    std::vector<yhh::WideNode> wide;
    I.depth = yhh::collapse_wide(tree, wide);
    lap("collapse to 4-wide");
    {  // depths of the 8- and 16-wide collapses (built at first use, ensure_wide_nodes): a wide node stands for every
       // internal binary node at a level that is a multiple of 3 (4), so the wide depth is 1 + deepest internal level / 3 (4)
      std::vector<int> level(tree.nodes.size(), 0);
      int deepest = 0;
      for (size_t n = 0; n < tree.nodes.size(); n++)
        if (tree.nodes[n].internal) {
          deepest = std::max(deepest, level[n]);
          level[(size_t)tree.nodes[n].start] = level[(size_t)tree.nodes[n].start + 1] = level[n] + 1;
        }
      I.depth8 = 1 + deepest / 3, I.depth16 = 1 + deepest / 4;
      I.node8_base = I.node16_base = 0;
    }
    I.root = tree.nodes[0].bbox, I.num_nodes = (int)wide.size(), I.num_prims = nel;
    {
      size_t at = nodes.size();
      nodes.resize(at + wide.size() * 8);
      memcpy(&nodes[at], wide.data(), wide.size() * sizeof(yhh::WideNode));
    }
    auto nrm = [&](int v) { return s.normals ? ld3(s.normals + 3 * (size_t)v) : F3{0, 0, 0}; };
    {  // leaf-ordered records (yh_device.h), filled in parallel
      const size_t per = lines ? 4 : 6, at = prims.size();
      prims.resize(at + per * (size_t)nel);
      yhd_float4* out = prims.data() + at;
      parallel_for(nel, [&](int slot) {
        int   e = tree.primitives[slot];
        float ew;
        memcpy(&ew, &e, 4);
        yhd_float4* r = out + per * (size_t)slot;
        if (lines) {
          int a = idx[2 * e], b = idx[2 * e + 1];
          F3  p0 = pos(a), p1 = pos(b), t0 = nrm(a), t1 = nrm(b);
          r[0] = {p0.x, p0.y, p0.z, rad(a)}, r[1] = {p1.x, p1.y, p1.z, rad(b)};
          r[2] = {t0.x, t0.y, t0.z, ew}, r[3] = {t1.x, t1.y, t1.z, 0};
        } else {
          int a = idx[3 * e], b = idx[3 * e + 1], cc = idx[3 * e + 2];
          F3  p0 = pos(a), p1 = pos(b), p2 = pos(cc), n0 = nrm(a), n1 = nrm(b), n2 = nrm(cc);
          r[0] = {p0.x, p0.y, p0.z, ew}, r[1] = {p1.x, p1.y, p1.z, 0}, r[2] = {p2.x, p2.y, p2.z, 0};
          r[3] = {n0.x, n0.y, n0.z, 0}, r[4] = {n1.x, n1.y, n1.z, 0}, r[5] = {n2.x, n2.y, n2.z, 0};
        }
      });
    }
    {
      // Per-vertex positions and per-element indices are read on the device only to sample a point
      // on an area light (triangles, pt.cpp:1287-1292) and to interpolate texture coordinates; the
      // traversal and the shading of a hit use the leaf records. Hair without texture coordinates —
      // nearly all of a scene's bytes — therefore has no entry in these arrays.
      const bool per_vertex = !lines || s.texcoords != nullptr;
      if (!per_vertex) {
        I.vert_base = 0, I.elem_base = 0;
      } else {
        const size_t at = vpos.size();
        vpos.resize(at + (size_t)s.num_vertices);
        vtex.resize(2 * (at + (size_t)s.num_vertices), 0.0f);
        if (s.texcoords) memcpy(&vtex[2 * at], s.texcoords, sizeof(float) * 2 * (size_t)s.num_vertices);
        parallel_for(s.num_vertices, [&](int v) {
          F3 p = pos(v);
          vpos[at + (size_t)v] = {p.x, p.y, p.z, lines ? rad(v) : 0.0f};
        });
        const size_t ea = elems.size();
        elems.resize(ea + (size_t)nel);
        parallel_for(nel, [&](int e) {
          elems[ea + (size_t)e] = lines ? yhd_int4{idx[2 * e], idx[2 * e + 1], 0, 0}
                                        : yhd_int4{idx[3 * e], idx[3 * e + 1], idx[3 * e + 2], 0};
        });
      }
    }
    if (lines && s.num_lines > best_lines) best_lines = s.num_lines, best_shape = si;
    lap("leaf records + vertex arrays");
    ctx->host_trees[(size_t)si] = std::move(tree);
  }
  // ---- layout of the one-lane kernels' copy of the trees (yh_device.h: lane_blob): test records first, nodes behind ----
  ctx->lane_shapes.assign((size_t)sd->num_shapes, yh_context::LaneShape{});
  {
    long long at = 0;
    for (int si = 0; si < sd->num_shapes; si++) {
      auto& L = ctx->lane_shapes[(size_t)si];
      L.kind = info[si].kind, L.node_base = info[si].node_base, L.num_nodes = info[si].num_nodes, L.prim_base = info[si].prim_base, L.num_prims = info[si].num_prims;
      L.test_off = at, at += (long long)L.num_prims * (L.kind == YH_KIND_LINES ? 1 : 2);
    }
    at = (at + 3) / 4 * 4 + 4;  // (nodes on 128-byte lines; four units of slack behind the last test record: a leaf step reads 64 bytes)
    if (at >= (1ll << 27)) return fail(ctx, YH_E_INVALID, "scene too large for 27-bit leaf offsets (%lld test-record units)", at);
    for (int si = 0; si < sd->num_shapes; si++) ctx->lane_shapes[(size_t)si].node_off = at, at += 4ll * info[si].num_nodes;
    if (at >= (1ll << 30)) return fail(ctx, YH_E_INVALID, "scene too large for 30-bit node offsets (%lld units)", at);
    ctx->lane_units = at + 4;
  }
  ctx->d_lane_blob.reset();
  // ---- objects and the scene-level BVH (pt.cpp:792-814) -------------------
  std::vector<yhd_object> objects(sd->num_objects);
  std::vector<yhh::Box>   obj_boxes(sd->num_objects);
  for (int oi = 0; oi < sd->num_objects; oi++) {
    auto& o = sd->objects[oi];
    if (o.shape < 0 || o.shape >= sd->num_shapes || o.material < 0 || o.material >= sd->num_materials)
      return fail(ctx, YH_E_INVALID, "object %d references a missing shape or material", oi);
    auto& I = info[o.shape];
    auto& d = objects[oi];
    memcpy(d.frame, o.frame, 48);
    inverse_frame(o.frame, true, d.inv_frame);
    d.kind = I.kind, d.node_base = I.node_base, d.prim_base = I.prim_base, d.vert_base = I.vert_base;
    d.elem_base = I.elem_base, d.has_normals = I.has_normals, d.material = o.material, d.has_texcoords = sd->shapes[o.shape].texcoords != nullptr;
    d.lane_root = (int)ctx->lane_shapes[(size_t)o.shape].node_off, d.lane_test = (int)ctx->lane_shapes[(size_t)o.shape].test_off, d.lane_pad0 = d.lane_pad1 = 0;
    // transform_bbox (math.h:3174-3185)
    const yhh::Box& b = I.root;
    float lo[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float hi[3] = {std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(),
        std::numeric_limits<float>::lowest()};
    for (int c = 0; c < 8; c++) {
      F3 corner = {(c & 4) ? b.max[0] : b.min[0], (c & 2) ? b.max[1] : b.min[1], (c & 1) ? b.max[2] : b.min[2]};
      F3 t      = transform_point(o.frame, corner);
      float tv[3] = {t.x, t.y, t.z};
      for (int k = 0; k < 3; k++) lo[k] = fmin_(lo[k], tv[k]), hi[k] = fmax_(hi[k], tv[k]);
    }
    for (int k = 0; k < 3; k++) obj_boxes[oi].min[k] = lo[k], obj_boxes[oi].max[k] = hi[k];
    {  // the same box with a margin a thousand times the rounding of either box test
      float ext = fmax_(fmax_(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
      float eps = 1e-3f * ext + 1e-5f;
      for (int k = 0; k < 3; k++) d.wbox_min[k] = lo[k] - eps, d.wbox_max[k] = hi[k] + eps;
      d.wbox_min[3] = d.wbox_max[3] = 0;  // (int bits) the shape's first 8- / 16-wide node once those arrays exist (ensure_wide_nodes)
    }
  }
  // array offsets on the device are 32-bit float4 indices
  if (prims.size() > (size_t)std::numeric_limits<int>::max() || nodes.size() > (size_t)std::numeric_limits<int>::max() ||
      vpos.size() > (size_t)std::numeric_limits<int>::max())
    return fail(ctx, YH_E_INVALID, "scene too large for 32-bit record offsets (%zu primitive, %zu node float4)", prims.size(), nodes.size());
  yhh::Tree scene_tree;
  yhh::build_bvh(scene_tree, obj_boxes);
  std::vector<yhd_float4> scene_nodes;
  for (auto& n : scene_tree.nodes) scene_nodes.push_back(node_lo(n)), scene_nodes.push_back(node_hi(n));
  int max_shape_depth = 0;
  for (auto& I : info) max_shape_depth = std::max(max_shape_depth, I.depth);
  // a wide node pushes at most three entries and keeps the fourth in a register
  ctx->stack_need = scene_tree.max_depth + 4 + 3 * max_shape_depth + 2;
  int max_shape_depth8 = 0;
  for (auto& I : info) max_shape_depth8 = std::max(max_shape_depth8, I.depth8);
  ctx->stack_need8 = scene_tree.max_depth + 4 + 7 * max_shape_depth8 + 2;  // an 8-wide node pushes at most seven
  int max_shape_depth16 = 0;
  for (auto& I : info) max_shape_depth16 = std::max(max_shape_depth16, I.depth16);
  ctx->stack_need16 = scene_tree.max_depth + 4 + 15 * max_shape_depth16 + 2;
  if (ctx->stack_need > yhk_stack_entries())
    return fail(ctx, YH_E_INVALID, "BVH too deep for the traversal stack (%d > %d)", ctx->stack_need, yhk_stack_entries());
  // ---- materials ---------------------------------------------------------
  std::vector<yhd_material> materials(sd->num_materials);
  int general_materials = 0;
  for (int i = 0; i < sd->num_materials; i++) {
    make_material(sd->materials[i], materials[i]);
    if (!materials[i].plain) general_materials = 1;
  }
  // ---- lights (pt.cpp:1695-1740) -----------------------------------------
  yhd_scene sc{};
  std::vector<float>      light_cdf;
  std::vector<yhd_float4> env_texels;
  std::vector<int>        small_lights;  // lights whose record goes into the LDS light table
  for (int oi = 0; oi < sd->num_objects; oi++) {
    auto& o = sd->objects[oi];
    auto& m = sd->materials[o.material];
    if (m.emission[0] == 0 && m.emission[1] == 0 && m.emission[2] == 0) continue;
    auto& s = sd->shapes[o.shape];
    if (s.num_lines > 0 || s.num_triangles <= 0) continue;
    if (sc.num_lights >= YH_MAX_LIGHTS) return fail(ctx, YH_E_INVALID, "more than %d lights", YH_MAX_LIGHTS);
    auto& L = sc.lights[sc.num_lights++];
    L.object = oi, L.environment = -1, L.cdf_base = (int)light_cdf.size(), L.cdf_count = s.num_triangles, L.small_base = -1;
    if (s.num_triangles <= YH_SMALL_LIGHT_TRIS) small_lights.push_back(sc.num_lights - 1);  // its record is made below, once the cdf exists
    else general_materials = 1;  // a light sampled and intersected through memory: the general kernel variant (dev_path.h: BIG_LIGHTS)
    for (int t = 0; t < s.num_triangles; t++) {
      F3 p0 = ld3(s.positions + 3 * (size_t)s.triangles[3 * t]), p1 = ld3(s.positions + 3 * (size_t)s.triangles[3 * t + 1]),
         p2 = ld3(s.positions + 3 * (size_t)s.triangles[3 * t + 2]);
      F3    c    = cross(p1 - p0, p2 - p0);
      float area = std::sqrt(dot(c, c)) / 2;  // triangle_area (math.h:3306)
      if (t) area += light_cdf.back();
      light_cdf.push_back(area);
    }
  }
  sc.num_environments = sd->num_environments;
  for (int ei = 0; ei < sd->num_environments; ei++) {
    auto& e = sd->environments[ei];
    auto& d = sc.environments[ei];
    memcpy(d.frame, e.frame, 48);
    inverse_frame(e.frame, false, d.inv_frame);
    memcpy(d.emission, e.emission, 12);
    d.tex_w = e.texels ? e.tex_width : 0, d.tex_h = e.texels ? e.tex_height : 0;
    d.texel_base = (int)env_texels.size();
    if (e.texels)
      for (size_t t = 0; t < (size_t)e.tex_width * e.tex_height; t++)
        env_texels.push_back({e.texels[3 * t], e.texels[3 * t + 1], e.texels[3 * t + 2], 0});
    if (e.emission[0] == 0 && e.emission[1] == 0 && e.emission[2] == 0) continue;
    if (sc.num_lights >= YH_MAX_LIGHTS) return fail(ctx, YH_E_INVALID, "more than %d lights", YH_MAX_LIGHTS);
    auto& L = sc.lights[sc.num_lights++];
    L.object = -1, L.environment = ei, L.cdf_base = (int)light_cdf.size(), L.cdf_count = 0, L.small_base = -1;
    if (e.texels) {
      size_t n    = (size_t)e.tex_width * e.tex_height;
      L.cdf_count = (int)n;
      for (size_t i = 0; i < n; i++) {
        int   iy    = (int)(i / e.tex_width);
        float th    = (iy + 0.5f) * pif / e.tex_height;
        float mx    = fmax_(fmax_(e.texels[3 * i], e.texels[3 * i + 1]), e.texels[3 * i + 2]);
        float value = mx * std::sin(th);
        if (i) value += light_cdf.back();
        light_cdf.push_back(value);
      }
    }
  }
  if (sc.num_lights == 0) return fail(ctx, YH_E_INVALID, "scene has no lights (the path sampler needs at least one)");
  // ---- tables the kernels keep in LDS (yh_device.h) -----------------------------------------------------
  // small area lights: root box, leaf-ordered triangles, area cdf — everything sample_lights / sample_lights_pdf read
  std::vector<yhd_float4> light_table;
  for (int li : small_lights) {
    auto& L  = sc.lights[li];
    auto& I  = info[sd->objects[L.object].shape];
    const yhd_float4* rec = prims.data() + I.prim_base;
    L.small_base = (int)light_table.size();
    yhd_float4 b0{I.root.min[0], I.root.min[1], I.root.min[2], 0}, b1{I.root.max[0], I.root.max[1], I.root.max[2], light_cdf[(size_t)L.cdf_base + L.cdf_count - 1]};
    memcpy(&b0.w, &L.cdf_count, 4);
    light_table.push_back(b0), light_table.push_back(b1);
    for (int t = 0; t < YH_SMALL_LIGHT_TRIS; t++)
      for (int k = 0; k < 3; k++) light_table.push_back(t < L.cdf_count ? rec[6 * t + k] : yhd_float4{0, 0, 0, 0});
    yhd_float4 cdf{0, 0, 0, 0};
    for (int t = 0; t < L.cdf_count; t++) (&cdf.x)[t] = light_cdf[(size_t)L.cdf_base + t];
    light_table.push_back(cdf);
  }
  // coarse index of the first textured environment light's cdf: 2048 entries (8 KB) halve the dependent fetches of
  // its 21-step binary search
  std::vector<float> env_tab;
  sc.env_tab_light = -1, sc.env_tab_k = 0, sc.env_tab_stride = 0;
  for (int li = 0; li < sc.num_lights && sc.env_tab_light < 0; li++) {
    auto& L = sc.lights[li];
    if (L.environment < 0 || L.cdf_count < 4096) continue;
    const int n = L.cdf_count, S = (n + 2047) / 2048, K = (n + S - 1) / S;
    env_tab.resize((size_t)K);
    for (int k = 0; k < K; k++) env_tab[(size_t)k] = light_cdf[(size_t)L.cdf_base + (size_t)std::min<int64_t>(n, (int64_t)(k + 1) * S) - 1];
    sc.env_tab_light = li, sc.env_tab_k = K, sc.env_tab_stride = S;
  }
  // ---- material colour textures (lookup_texture's per-texel conversion done once, pt.cpp:147-164) --------
  std::vector<yhd_texture> textures((size_t)std::max(0, sd->num_textures));
  std::vector<yhd_float4>  tex_texels;
  {
    std::vector<char> need_linear(textures.size(), 0);
    for (int i = 0; i < sd->num_materials; i++) {
      auto& m = sd->materials[i];
      for (int id : {m.emission_tex, m.color_tex, m.scattering_tex})
        if (id < 0 || id > sd->num_textures) return fail(ctx, YH_E_INVALID, "material %d references a missing texture", i);
      if (m.emission_tex > 0) need_linear[(size_t)m.emission_tex - 1] = 1;  // transmission *= emission_tex.x, linear (pt.cpp:421)
    }
    auto srgb_to_rgb = [](float srgb) {  // math.h:3742-3745
      return (srgb <= 0.04045) ? srgb / 12.92f : std::pow((srgb + 0.055f) / (1.0f + 0.055f), 2.4f);
    };
    for (size_t t = 0; t < textures.size(); t++) {
      auto& src = sd->textures[t];
      if (src.width <= 0 || src.height <= 0 || !src.pixels) return fail(ctx, YH_E_INVALID, "texture %d is empty", (int)t);
      size_t n = (size_t)src.width * src.height;
      auto&  d = textures[t];
      d.width = src.width, d.height = src.height, d.srgb_base = (int)tex_texels.size(), d.linear_base = -1;
      tex_texels.resize(tex_texels.size() + n);
      yhd_float4* out = tex_texels.data() + d.srgb_base;
      if (src.is_byte) {
        auto b = (const unsigned char*)src.pixels;
        parallel_for((int)n, [&](int i) {
          out[i] = {srgb_to_rgb(b[3 * (size_t)i] / 255.0f), srgb_to_rgb(b[3 * (size_t)i + 1] / 255.0f),
              srgb_to_rgb(b[3 * (size_t)i + 2] / 255.0f), 0};
        });
        if (need_linear[t]) {
          d.linear_base = (int)tex_texels.size();
          tex_texels.resize(tex_texels.size() + n);
          yhd_float4* lin = tex_texels.data() + d.linear_base;
          parallel_for((int)n, [&](int i) { lin[i] = {b[3 * (size_t)i] / 255.0f, b[3 * (size_t)i + 1] / 255.0f, b[3 * (size_t)i + 2] / 255.0f, 0}; });
        }
      } else {
        auto f = (const float*)src.pixels;
        parallel_for((int)n, [&](int i) { out[i] = {f[3 * (size_t)i], f[3 * (size_t)i + 1], f[3 * (size_t)i + 2], 0}; });
        d.linear_base = d.srgb_base;
      }
    }
  }
  lap("objects, materials, lights");
  // ---- upload ------------------------------------------------------------
  int rc;
  if ((rc = upload(ctx, ctx->d_nodes, nodes.data(), nodes.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_prims, prims.data(), prims.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_vpos, vpos.data(), vpos.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_elems, elems.data(), elems.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_objects, objects.data(), objects.size() * sizeof(yhd_object)))) return rc;
  ctx->host_objects = objects;
  ctx->object_shape.resize((size_t)sd->num_objects);
  for (int oi = 0; oi < sd->num_objects; oi++) ctx->object_shape[(size_t)oi] = sd->objects[oi].shape;
  if ((rc = upload(ctx, ctx->d_materials, materials.data(), materials.size() * sizeof(yhd_material)))) return rc;
  if ((rc = upload(ctx, ctx->d_scene_nodes, scene_nodes.data(), scene_nodes.size() * 16))) return rc;
  std::vector<int> scene_prims_padded = scene_tree.primitives;
  scene_prims_padded.resize((scene_prims_padded.size() + 3) / 4 * 4, 0);  // staged to LDS as float4
  if ((rc = upload(ctx, ctx->d_scene_prims, scene_prims_padded.data(), scene_prims_padded.size() * 4))) return rc;
  if ((rc = upload(ctx, ctx->d_light_cdf, light_cdf.data(), light_cdf.size() * 4))) return rc;
  if ((rc = upload(ctx, ctx->d_light_table, light_table.data(), light_table.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_env_tab, env_tab.data(), env_tab.size() * 4))) return rc;
  if ((rc = upload(ctx, ctx->d_env_texels, env_texels.data(), env_texels.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_textures, textures.data(), textures.size() * sizeof(yhd_texture)))) return rc;
  if ((rc = upload(ctx, ctx->d_tex_texels, tex_texels.data(), tex_texels.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_vtex, vtex.data(), vtex.size() * 4))) return rc;
  lap("hipMalloc + H2D copies");
  sc.nodes = (const yhd_float4*)ctx->d_nodes.p, sc.prims = (const yhd_float4*)ctx->d_prims.p;
  sc.vpos = (const yhd_float4*)ctx->d_vpos.p;
  sc.elems = (const yhd_int4*)ctx->d_elems.p;
  sc.objects = (const yhd_object*)ctx->d_objects.p, sc.materials = (const yhd_material*)ctx->d_materials.p;
  sc.scene_nodes = (const yhd_float4*)ctx->d_scene_nodes.p, sc.scene_prims = (const int*)ctx->d_scene_prims.p;
  sc.num_scene_nodes = (int)scene_tree.nodes.size(), sc.num_objects = sd->num_objects;
  sc.light_cdf = (const float*)ctx->d_light_cdf.p, sc.env_texels = (const yhd_float4*)ctx->d_env_texels.p;
  sc.light_table = (const yhd_float4*)ctx->d_light_table.p, sc.light_table_f4 = (int)light_table.size();
  sc.env_tab = (const float*)ctx->d_env_tab.p;
  sc.stack_entries = std::max(8, (ctx->stack_need + 7) / 8 * 8);
  sc.nodes8 = nullptr, sc.num_nodes8_total = 0;  // built at first use: ensure_wide_nodes
  sc.stack_entries8 = std::max(8, (ctx->stack_need8 + 7) / 8 * 8);
  sc.nodes16 = nullptr, sc.num_nodes16_total = 0;
  sc.stack_entries16 = std::max(8, (ctx->stack_need16 + 7) / 8 * 8);
  sc.lane_blob = nullptr, sc.lane_blob_units = 0;  // filled at first use: ensure_lane_blob
  sc.textures = (const yhd_texture*)ctx->d_textures.p, sc.tex_texels = (const yhd_float4*)ctx->d_tex_texels.p;
  sc.vtex = (const float*)ctx->d_vtex.p;
  memcpy(sc.camera.frame, sd->camera.frame, 48);
  sc.camera.lens = sd->camera.lens, sc.camera.film_x = sd->camera.film[0], sc.camera.film_y = sd->camera.film[1];
  sc.camera.focus = sd->camera.focus, sc.camera.aperture = sd->camera.aperture;
  sc.num_nodes_total = (int)(nodes.size() / 8), sc.num_prim_f4 = (int)prims.size();
  // nodelets: the top (breadth-first prefix) of the largest hair shape's BVH
  sc.general_materials = general_materials;
  {  // scene-level LDS table: objects (8 float4 each), scene BVH nodes (2 float4 each), primitive ids
    static_assert(sizeof(yhd_object) == 16 * YH_OBJECT_F4, "yhd_object is staged to LDS as float4");
    int f4 = YH_OBJECT_F4 * sd->num_objects + 2 * (int)scene_tree.nodes.size() + (sd->num_objects + 3) / 4;
    sc.lds_scene_f4 = f4 * 16 <= 8192 ? f4 : 0;
  }
  {  // the material table in LDS; the plain kernel variants rely on it and on the scene-level table (dev_path.h)
    static_assert(sizeof(yhd_material) == 16 * YH_MATERIAL_F4, "yhd_material is staged to LDS as float4");
    sc.lds_materials = sd->num_materials <= 24 ? sd->num_materials : 0;
    if (sc.lds_materials == 0 || sc.lds_scene_f4 == 0) sc.general_materials = 1;
  }
  sc.lds_node_base = 0, sc.lds_node_count = 0;
  if (best_shape >= 0) {
    sc.lds_node_base  = info[best_shape].node_base;
    int want = 0;  // LDS nodelets are optional (YHAIR_LDS_NODES): measured no gain once a step is a single fetch, see DESIGN.md
#if YH_LDS_NODELETS
    if (const char* env = getenv("YHAIR_LDS_NODES")) want = std::max(0, atoi(env));
#endif
    // 128 B per nodelet next to the stacks and the scene table of the larger launch shape: stay inside the CU's 160 KB
    sc.lds_node_count = 0;
    int room = (160 * 1024 - yhk_trace_lds_bytes(&sc, 0)) / 128;
    sc.lds_node_count = std::max(0, std::min({info[best_shape].num_nodes, want, room}));
  }
  ctx->scene      = sc;
  {  // fingerprint of the scene for the process-wide trial record: counts, camera, materials, objects, a sample of the geometry
    uint64_t h = 1469598103934665603ULL;
    auto mix = [&](const void* p, size_t n) {
      const unsigned char* b = (const unsigned char*)p;
      for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ULL;
    };
    mix(&sd->camera, sizeof(sd->camera));
    mix(sd->materials, sizeof(yh_material) * (size_t)sd->num_materials);
    mix(sd->objects, sizeof(yh_object) * (size_t)sd->num_objects);
    for (int i = 0; i < sd->num_shapes; i++) {
      const yh_shape& sh = sd->shapes[i];
      int counts[3] = {sh.num_vertices, sh.num_lines, sh.num_triangles};
      mix(counts, sizeof(counts));
      if (sh.positions && sh.num_vertices > 0) {
        const size_t n = (size_t)sh.num_vertices, take = std::min<size_t>(n, 256);
        mix(sh.positions, take * 12), mix(sh.positions + 3 * (n - take), take * 12);
      }
    }
    for (int i = 0; i < sd->num_environments; i++) mix(&sd->environments[i], offsetof(yh_environment, texels));
    ctx->scene_key = h;
  }
  ctx->d_scene_copy.reset();  // (stream_impl uploads the new table at its first launch)
  ctx->have_scene = true;
  ctx->have_state = false;
  ctx->launch_shape = 0;  // a new scene: no measured costs yet
  ctx->item_cost.clear();
  ctx->have_costs = false, ctx->costs_settled = false, ctx->dense = -1, ctx->chain = -1, ctx->chain16 = -1;
  for (double& t : ctx->shape_ms) t = 0;
  for (int& t : ctx->shape_trials) t = 0;
  wide_build_start(ctx);  // the wide collapses in the background: ready by the time a kernel that needs them is tried
  ctx->trials_from_disk = false;
  // the kernels over 4-wide nodes — quads and one lane per path alike — read the trees from the lane blob (yh_device.h): made
  // here, on the device, from the arrays just uploaded (two streaming kernels per shape, about a millisecond)
  return ensure_lane_blob(ctx);
}

int yh_set_shard(yh_context* ctx, int rank, int world) {
  if (!ctx) return YH_E_INVALID;
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, YH_E_INVALID, "bad shard %d of %d", rank, world);
  if (rank != ctx->rank || world != ctx->world) ctx->item_cost.clear();  // another shard is another image to plan and to time kernels on: yh_init_state starts over
  ctx->rank = rank, ctx->world = world;
  ctx->have_state = false;
  return YH_OK;
}

int yh_init_state(yh_context* ctx, const yh_trace_params* params) {
  if (!ctx) return YH_E_INVALID;
  if (!ctx->have_scene) return fail(ctx, YH_E_STATE, "yh_init_state before yh_upload_scene");
  if (!params || params->resolution <= 0 || params->bounces < 0)
    return fail(ctx, YH_E_INVALID, "bad trace params");
  if (params->shader < 0 || params->shader >= YH_SHADER_COUNT)
    return fail(ctx, YH_E_INVALID, "sampler unknown");  // get_trace_shader_func's throw (pt.cpp:1669)
  if (params->hair_exact && params->shader != YH_SHADER_PATH) return fail(ctx, YH_E_INVALID, "hair_exact exists for the path shader only");
  HIPCHK(ctx, hipSetDevice(ctx->device));
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
  int tx = tiles_of(w), ty = tiles_of(h);
  ctx->num_tiles_total = tx * ty;
  auto& owned = ctx->owned;
  owned.clear();
  for (int t = ctx->rank; t < ctx->num_tiles_total; t += ctx->world) owned.push_back(t);
  bool new_image = false;
  if ((int)ctx->item_cost.size() != ctx->num_tiles_total * 4) {  // scheduling hints survive a re-init of the same image
    ctx->item_cost.assign((size_t)ctx->num_tiles_total * 4, 0);
    ctx->have_costs = false, ctx->costs_settled = false, ctx->dense = -1, ctx->chain = -1, ctx->chain16 = -1, ctx->launch_shape = 0;
    for (double& t : ctx->shape_ms) t = 0;
    for (int& t : ctx->shape_trials) t = 0;
    new_image = true;
  }
  std::vector<int> tiles;
  build_work_items(ctx, tiles);
  const int first_shape = getenv("YHAIR_SHAPE") ? choose_launch_shape(ctx) : ctx->launch_shape;
  ctx->state.tiles_x = tx, ctx->state.num_groups = 1, ctx->state.group_begin[0] = 0, ctx->state.group_begin[1] = (int)tiles.size();
  ctx->state.static_items = 0;
  if (params->shader == YH_SHADER_PATH && first_shape == 3) deal_items_for_stream(ctx, tiles);
  if (params->shader == YH_SHADER_PATH && (first_shape == 4 || first_shape == 7)) split_items_for_octets(tiles);
  if (params->shader == YH_SHADER_PATH && first_shape == 5) split_items_side_by_side(ctx, tiles);
  if (params->shader == YH_SHADER_PATH && (first_shape == 6 || first_shape == 8)) split_items_for_hex(tiles);
  if (!getenv("YHAIR_NO_LAYOUT")) lay_out_first_round(ctx, tiles, params->shader == YH_SHADER_PATH ? first_shape : 0);  // (developer switch: the plain cost order)
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
  ctx->launches_of_state = 0;
  s.samples_done = 0, s.bounces = params->bounces, s.clamp = params->clamp, s.shader = params->shader;
  s.shard_rank = ctx->rank, s.shard_world = ctx->world;
  ctx->have_state = true;
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
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->state.samples_done = 0, ctx->launches_of_state = 0, ctx->last_shape = -1, ctx->last_ms = 0, ctx->last_launches = 0;
    ctx->have_costs = true;  // the launches that follow are planned: their times rank the kernels
  }
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
static int upload_work_items(yh_context* ctx) {
  std::vector<int> tiles;
  build_work_items(ctx, tiles);
  ctx->state.num_groups = 1, ctx->state.group_begin[0] = 0, ctx->state.group_begin[1] = (int)tiles.size();
  ctx->state.static_items = 0;
  ctx->state.prio_items   = getenv("YHAIR_PRIO_ITEMS") ? atoi(getenv("YHAIR_PRIO_ITEMS")) : 0;  // developer A/B switch (a library built with -DYH_LAB_PRIO)
  if (ctx->state.shader == YH_SHADER_PATH && ctx->state.launch_shape == 3) deal_items_for_stream(ctx, tiles);
  if (ctx->state.shader == YH_SHADER_PATH && (ctx->state.launch_shape == 4 || ctx->state.launch_shape == 7)) split_items_for_octets(tiles);
  if (ctx->state.shader == YH_SHADER_PATH && ctx->state.launch_shape == 5) split_items_side_by_side(ctx, tiles);
  if (ctx->state.shader == YH_SHADER_PATH && (ctx->state.launch_shape == 6 || ctx->state.launch_shape == 8)) split_items_for_hex(tiles);
  if (!getenv("YHAIR_NO_LAYOUT")) lay_out_first_round(ctx, tiles, ctx->state.shader == YH_SHADER_PATH ? ctx->state.launch_shape : 0);
  ctx->state.num_tiles = (int)tiles.size();
  HIPCHK(ctx, hipMemcpy(ctx->d_tiles.p, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
  return YH_OK;
}

// SIDE BY SIDE (launch shape 5). The launch of a sparse image ends with its most expensive items: every expensive item runs
// from the start, and the launch is as long as the longest chain (C1: the most expensive quadrant takes 14.8 ms per 64
// samples, the median expensive one 9 ms). The same handful of quadrants tops EVERY launch, and the octet form runs an item
// in 0.74 x the time for two waves instead of one — so the first K items of the cost-sorted list run as octets and
// everything else as quads, in ONE launch (csrc/kernels.hip: k_trace_sbs): its first workgroups take the octet entries,
// the others the quad items. The first workgroups of a launch get the fastest wave slots (lay_out_first_round below), the
// workgroups are of one size, and there is one dispatch order — the three things the earlier forms of this idea lacked
// (two kernels on two streams: the streams raced for the slots and the workgroup sizes did not pack, 14.8 -> 18.7 ms;
// one kernel whose waves pick the form per item: 5-10 % behind before any item was widened). Both forms render the quad
// kernel's bits; each pixel belongs to one of them. MEASURED (profiles/r03/side_by_side_fused_ab.txt): C1 at 720^2 14.9 ->
// 13.4 ms per 64 samples with 16-128 items widened (4: 14.5, 512: 14.7, 1024: 16.2; 0, the control: 15.4), the bench
// 2 182 -> 2 457 Msamples/s. A trial candidate on sparse images that are not chain-bound (on those the wider kernels run
// every item wide).
// Expensive = within 5 x of the most expensive item. Returns how many of them there are.
static int expensive_items(const yh_context* ctx, const std::vector<int>& items) {
  if (items.empty()) return 0;
  const uint64_t top = ctx->item_cost[(size_t)items[0]];
  int n = 0;
  for (int it : items) {  // (cost-sorted as far as the expensive ones go)
    if ((uint64_t)ctx->item_cost[(size_t)it] * 5 < top || top == 0) break;
    n++;
  }
  return n;
}
// The side-by-side launch's workgroups: octet ones first (eight waves each, one half-quadrant entry per wave at a time), quad ones behind.
static bool side_by_side_grids(const yh_context* ctx, int* oct_blocks, int* quad_blocks) {
  const int lds = yhk_trace_sbs_lds_bytes(&ctx->scene), occ = yhk_trace_sbs_occupancy(lds, ctx->scene.general_materials);
  if (occ < 1) return false;
  const int resident = ctx->num_cus * occ;
  *oct_blocks  = std::min((ctx->hy_oct_entries + 7) / 8, resident / 2);  // (the quad workgroups keep at least half of the device, whatever YHAIR_HY_OCT says)
  *quad_blocks = ctx->hy_quad_items > 0 ? std::max(1, std::min((ctx->hy_quad_items + 7) / 8, resident - *oct_blocks)) : 0;
  return true;
}
static void split_items_side_by_side(yh_context* ctx, std::vector<int>& items) {
  // How many: the launch of a sparse image ends with a handful of quadrants that are the most expensive ones in EVERY launch
  // (C1: sixteen widened items take 10 % off the launch, 128 no more, 512 lose it again to the extra waves —
  // profiles/r03/side_by_side_fused_ab.txt): a sixty-fourth of the expensive items, sixteen at least.
  const int H = expensive_items(ctx, items);
  int n_oct = std::min(H / 2, std::max(16, std::min(256, H / 64)));
  if (const char* env = getenv("YHAIR_HY_OCT")) n_oct = std::max(0, std::min((int)items.size(), atoi(env)));  // developer switch
  std::vector<int> out;
  out.reserve(items.size() + n_oct);
  for (size_t i = (size_t)n_oct; i < items.size(); i++) out.push_back(items[i]);                   // quads: the rest, most expensive first
  for (int i = 0; i < n_oct; i++) out.push_back(items[i] << 1), out.push_back((items[i] << 1) | 1);  // octets: two half-quadrant entries each
  ctx->hy_quad_items = (int)items.size() - n_oct, ctx->hy_oct_entries = 2 * n_oct;
  ctx->hy_oct_items.assign(items.begin(), items.begin() + n_oct);
  items.swap(out);
  if (!getenv("YHAIR_NO_LAYOUT")) {  // both lists by wave slot: the octet workgroups are the first of the launch, the quad ones follow (side_by_side_impl)
    int G_o = 0, G_q = 0;
    side_by_side_grids(ctx, &G_o, &G_q);
    if (G_o > 0) lay_out_range(ctx, items.data() + ctx->hy_quad_items, (size_t)ctx->hy_oct_entries, 8, G_o, 0);
    if (G_q > 0) lay_out_range(ctx, items.data(), (size_t)ctx->hy_quad_items, 8, G_q, G_o);
  }
}
static int side_by_side_impl(yh_context* ctx, int nsamples, bool sync);

static void wide_build_join(yh_context* ctx) {
  if (ctx->wide_job && ctx->wide_job->th.joinable()) ctx->wide_job->th.join();
}
static void wide_build_start(yh_context* ctx) {  // (ctx->host_trees must stay untouched until wide_build_join)
  wide_build_join(ctx);
  ctx->wide_job.reset(new WideBuild());
  WideBuild*                    job   = ctx->wide_job.get();
  const std::vector<yhh::Tree>* trees = &ctx->host_trees;
  const size_t                  ns    = trees->size();
  job->w8.resize(ns), job->w16.resize(ns), job->d8.assign(ns, 0), job->d16.assign(ns, 0);
  job->th = std::thread([job, trees, ns] {
    std::vector<std::thread> pool;
    for (size_t si = 0; si < ns; si++) {
      pool.emplace_back([job, trees, si] { job->d8[si] = yhh::collapse_wide8((*trees)[si], job->w8[si]); });
      pool.emplace_back([job, trees, si] { job->d16[si] = yhh::collapse_wide16((*trees)[si], job->w16[si]); });
      if (pool.size() >= 8) {
        for (auto& t : pool) t.join();
        pool.clear();
      }
    }
    for (auto& t : pool) t.join();
  });
}
// The 8- and 16-wide collapses of the shapes' trees (host/bvh_build.h), built, uploaded and wired into the object records
// when a kernel that traverses them is about to run for the first time (launch shapes 4, 5, 6, 7).
static int ensure_wide_nodes(yh_context* ctx) {
  if (ctx->wide_built) return YH_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (a queued launch may be reading the object records)
  if (!ctx->wide_job) wide_build_start(ctx);  // (normally started by yh_upload_scene)
  wide_build_join(ctx);
  const size_t ns = ctx->host_trees.size();
  std::vector<std::vector<yhh::WideNode8>>&  w8  = ctx->wide_job->w8;
  std::vector<std::vector<yhh::WideNode16>>& w16 = ctx->wide_job->w16;
  std::vector<int>&                          d8 = ctx->wide_job->d8, &d16 = ctx->wide_job->d16;
  {  // the LDS stacks were sized at upload from the depths these collapses were expected to have
    int m8 = 0, m16 = 0;
    for (size_t si = 0; si < ns; si++) m8 = std::max(m8, d8[si]), m16 = std::max(m16, d16[si]);
    if (7 * m8 > ctx->stack_need8 || 15 * m16 > ctx->stack_need16)
      return fail(ctx, YH_E_INVALID, "wide trees deeper than their traversal stacks were sized for (%d / %d levels)", m8, m16);
  }
  std::vector<int>        base8(ns), base16(ns);
  std::vector<yhd_float4> nodes8, nodes16;
  for (size_t si = 0; si < ns; si++) {
    base8[si] = (int)(nodes8.size() / 16), base16[si] = (int)(nodes16.size() / 32);
    size_t at = nodes8.size();
    nodes8.resize(at + w8[si].size() * 16);
    if (!w8[si].empty()) memcpy(&nodes8[at], w8[si].data(), w8[si].size() * sizeof(yhh::WideNode8));
    at = nodes16.size();
    nodes16.resize(at + w16[si].size() * 32);
    if (!w16[si].empty()) memcpy(&nodes16[at], w16[si].data(), w16[si].size() * sizeof(yhh::WideNode16));
  }
  if (nodes8.size() > (size_t)std::numeric_limits<int>::max() || nodes16.size() > (size_t)std::numeric_limits<int>::max())
    return fail(ctx, YH_E_INVALID, "scene too large for 32-bit wide-node offsets");
  int rc;
  if ((rc = upload(ctx, ctx->d_nodes8, nodes8.data(), nodes8.size() * 16))) return rc;
  if ((rc = upload(ctx, ctx->d_nodes16, nodes16.data(), nodes16.size() * 16))) return rc;
  for (size_t oi = 0; oi < ctx->host_objects.size(); oi++) {
    const size_t si = (size_t)ctx->object_shape[oi];
    memcpy(&ctx->host_objects[oi].wbox_min[3], &base8[si], 4);
    memcpy(&ctx->host_objects[oi].wbox_max[3], &base16[si], 4);
  }
  HIPCHK(ctx, hipMemcpy(ctx->d_objects.p, ctx->host_objects.data(), ctx->host_objects.size() * sizeof(yhd_object), hipMemcpyHostToDevice));
  ctx->scene.nodes8 = (const yhd_float4*)ctx->d_nodes8.p, ctx->scene.num_nodes8_total = (int)(nodes8.size() / 16);
  ctx->scene.nodes16 = (const yhd_float4*)ctx->d_nodes16.p, ctx->scene.num_nodes16_total = (int)(nodes16.size() / 32);
  ctx->d_scene_copy.reset();  // (the copy of the scene table in device memory is made again at its next use)
  ctx->wide_built = true;
  ctx->wide_job.reset();
  ctx->host_trees.clear(), ctx->host_trees.shrink_to_fit();
  return YH_OK;
}

// THE HEAD OF THE LIST BY POSITION. A wave of k_trace takes its first item from the list entry at its own position
// (workgroup x waves per workgroup + wave; csrc/dev_items.h) and later ones from the cursor behind those positions. The four
// wave slots of a SIMD do not run at the same speed: on C1 the same kind of item takes 10.4 ms in hardware slot 0, 11.0 in
// slot 1, 11.8 in slot 2 and 13.3 in slot 3 (profiles/r03/where_items_ran.txt: the issue arbiter favours the older wave), and
// the launch ends with its slowest item. A wave's slot follows from the dispatch order: the workgroups come round by round,
// one per CU and round, and waves w and w + 4 of a 512-thread workgroup share a SIMD — so slot = round x (waves per workgroup
// / 4) + wave / 4. The most expensive items go to the slot-0 waves, the next to slot 1, and so on: on a sparse image the
// slowest slot holds none of the expensive items. Purely a matter of time: whatever the layout, every entry is taken once.
static void lay_out_range(const yh_context* ctx, int* items, size_t n, int wpb, int G, int block_offset) {  // entries [0, n) of one list, its G workgroups; block_offset: workgroups of the same launch dispatched before them
  const size_t P = std::min((size_t)G * wpb, n);  // entries taken by position
  std::vector<std::pair<uint64_t, uint32_t>> order;  // (slot class, place inside it) -> position
  order.reserve(P);
  for (size_t pos = 0; pos < P; pos++) {
    const uint64_t b = pos / wpb, w = pos % wpb;
    const uint64_t g = b + (uint64_t)block_offset, cls = (g / ctx->num_cus) * ((wpb + 3) / 4) + w / 4;
    order.emplace_back((cls << 40) | ((g % ctx->num_cus) << 8) | (w % 4), (uint32_t)pos);  // (which item shares a SIMD with which makes no difference: snake order measured equal)
  }
  std::sort(order.begin(), order.end());
  std::vector<int> head(P);
  for (size_t k = 0; k < P; k++) head[order[k].second] = items[k];  // the k-th most expensive item on the k-th fastest wave
  std::copy(head.begin(), head.end(), items);
}
static void lay_out_first_round(const yh_context* ctx, std::vector<int>& items, int shape) {
  if (shape == 3 || shape == 5 || items.empty()) return;  // (k_stream deals its items itself; side by side lays its two lists out when it splits them)
  const int wpb = yhk_block_threads(shape) / 64;
  const int occ = yhk_trace_occupancy(yhk_trace_lds_bytes(&ctx->scene, shape), ctx->scene.general_materials, shape);
  if (occ < 1 || wpb < 1) return;
  const int G = std::max(1, std::min(((int)items.size() + wpb - 1) / wpb, ctx->num_cus * occ));  // the grid trace_impl launches
  lay_out_range(ctx, items.data(), items.size(), wpb, G);
}

// Bookkeeping after a synchronous launch: its time (kernel selection) and, after launches 1, 2, 4, 8, ... of a state,
// the longest-processing-time-first order for the next ones (the pixel results do not depend on either).
static int replan_after_launch(yh_context* ctx, int nsamples) {
  // A pixel's samples are sequential, so the items that start last bound the launch; hair quadrants
  // cost 10-100x background ones. Re-planned after launches 1, 2, 4, 8, ... of a state: the relative
  // costs of the items settle after the first launches (they are a property of the image), and the
  // read-back, sort and upload are a few hundred microseconds of a 16 ms launch.
  const unsigned li      = ++ctx->launches_of_state;
  // (... and after the first launch long enough to settle the costs, whenever it comes: the kernel trials wait for it)
  const bool     refresh = (li & (li - 1)) == 0 || (!ctx->costs_settled && nsamples >= YH_TRIAL_SPP && ctx->state.shader == YH_SHADER_PATH);
  if (refresh) HIPCHK(ctx, hipMemcpy(ctx->item_cost.data(), ctx->d_tile_cost.p, ctx->item_cost.size() * 4, hipMemcpyDeviceToHost));
  if (refresh && ctx->last_shape == 5)  // an item that ran as octets reports the time of its two halves, 2 x 0.74 of what it costs as a quad
    for (int it : ctx->hy_oct_items) ctx->item_cost[(size_t)it] = (unsigned int)((double)ctx->item_cost[(size_t)it] * (1.0 / 1.48));
  if (ctx->state.shader == YH_SHADER_PATH && (refresh || ctx->have_costs)) record_launch(ctx, nsamples, refresh);
  if (!refresh) return YH_OK;
  return upload_work_items(ctx);
}

#ifdef YH_LAB_WAVEFRONT
// One launch of the wavefront integrator (csrc/lab/wavefront.hip): persistent workgroups, one path pool each.
static int wavefront_impl(yh_context* ctx, int nsamples, bool sync) {
  int k = 1;
  if (const char* env = getenv("YHAIR_WF_SLOTS")) k = atoi(env) >= 2 ? 2 : 1;  // path slots per thread (developer switch)
  const int P         = yhk_wavefront_slots(k);
  const int stack     = std::max(8, (ctx->stack_need + 7) / 8 * 8);
  const int lds_bytes = yhk_wavefront_lds_bytes(stack, YHD_LDS_TABLES_F4(&ctx->scene), k);
  const int occupancy = yhk_wavefront_occupancy(lds_bytes, ctx->scene.general_materials, k);
  if (occupancy < 1) return fail(ctx, YH_E_DEVICE, "k_wavefront cannot run with %d bytes of LDS per block", lds_bytes);
  const int64_t pixels = (int64_t)ctx->state.num_tiles * 16;  // work items are 4x4 pixel quadrants
  const int     grid   = (int)std::max<int64_t>(1, std::min<int64_t>((pixels + P - 1) / P, (int64_t)ctx->num_cus * occupancy));
  const size_t  slots  = (size_t)grid * P;
  if (slots > ctx->pool_slots) {
    int rc;
    if ((rc = alloc_zero(ctx, ctx->d_pool_ray_o, slots * 16)) || (rc = alloc_zero(ctx, ctx->d_pool_ray_d, slots * 16)) ||
        (rc = alloc_zero(ctx, ctx->d_pool_weight, slots * 16)) || (rc = alloc_zero(ctx, ctx->d_pool_radiance, slots * 16)) ||
        (rc = alloc_zero(ctx, ctx->d_pool_hit, slots * 16)))
      return rc;
    ctx->pool_slots = slots;
    ctx->pool.ray_o = (yhd_float4*)ctx->d_pool_ray_o.p, ctx->pool.ray_d = (yhd_float4*)ctx->d_pool_ray_d.p;
    ctx->pool.weight = (yhd_float4*)ctx->d_pool_weight.p, ctx->pool.radiance = (yhd_float4*)ctx->d_pool_radiance.p;
    ctx->pool.hit = (yhd_int4*)ctx->d_pool_hit.p;
  }
  // the medium of a path inside a volume: two float4 per slot, general scenes only (a plain scene's kernel never
  // touches it). Its capacity is tracked on its own: a context that rendered a plain scene first has none yet.
  if (ctx->scene.general_materials && slots > ctx->pool_medium_slots) {
    int rc;
    if ((rc = alloc_zero(ctx, ctx->d_pool_medium, slots * 32))) return rc;
    ctx->pool_medium_slots = slots;
    ctx->pool.medium       = (yhd_float4*)ctx->d_pool_medium.p;
  }
  ctx->pool.slots_per_block = P, ctx->pool.stack_entries = stack;
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cursor.p, 0, 8 * 16 * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cost.p, 0, (size_t)ctx->num_tiles_total * 16, ctx->stream));
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int e = yhk_wavefront(&ctx->scene, &ctx->state, nsamples, &ctx->pool, k, grid, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_wavefront launch: %s", hipGetErrorString((hipError_t)e));
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->state.samples_done += nsamples;
  ctx->last_launches = 1;
  if (sync) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->ev0, ctx->ev1));
    return replan_after_launch(ctx, nsamples);
  }
  return YH_OK;
}
#endif

// The one-lane kernels' copy of the shape trees (yh_device.h: yhd_scene::lane_blob), made on the device from the node and
// primitive arrays at the first launch that needs it: an image that never runs k_stream / k_intersect_lanes does not pay
// the memory (test records 32 B per segment + the nodes once more).
static int ensure_lane_blob(yh_context* ctx) {
  if (ctx->scene.lane_blob) return YH_OK;
  int rc;
  if ((rc = alloc_zero(ctx, ctx->d_lane_blob, (size_t)ctx->lane_units * 32))) return rc;
  for (auto& L : ctx->lane_shapes) {
    int e = yhk_lane_blob_shape(ctx->scene.nodes, ctx->scene.prims, (yhd_float4*)ctx->d_lane_blob.p, L.kind, L.node_base, L.num_nodes, L.prim_base,
        L.num_prims, L.node_off, L.test_off, ctx->stream);
    if (e) return fail(ctx, YH_E_DEVICE, "lane blob build: %s", hipGetErrorString((hipError_t)e));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->scene.lane_blob = (const yhd_float4*)ctx->d_lane_blob.p, ctx->scene.lane_blob_units = ctx->lane_units;
  ctx->d_scene_copy.reset();  // (the scene table in device memory is made again at its next use)
  return YH_OK;
}

// Launch geometry of the streaming integrator: path slots per wave and workgroups. The pixels of the launch are
// spread over as many waves as the CUs hold, each wave with a few paths per lane so that its lanes stay full
// between stages: 128 .. 192 slots (more waves beat fuller batches: measured on C2 / C3, profiles/r02;
// YHAIR_ST_SLOTS / YHAIR_ST_WAVES: developer switches). Returns 0 when the kernel cannot run.
static int stream_geometry(const yh_context* ctx, int num_items, int* slots_per_wave, int* grid_blocks, int* lds_out) {
  const int     wpb    = yhk_stream_block_threads() / 64;
  const int64_t pixels = (int64_t)num_items * 16;  // work items are 4x4 pixel quadrants
  int           P      = (int)std::max<int64_t>(128, std::min<int64_t>(192, (pixels / ((int64_t)ctx->num_cus * 16) + 63) / 64 * 64));
  if (const char* env = getenv("YHAIR_ST_SLOTS")) P = std::max(64, std::min(4096, atoi(env) / 64 * 64));
  const int lds_bytes = yhk_stream_lds_bytes(YHD_LDS_TABLES_F4(&ctx->scene), P);
  int       occupancy = yhk_stream_occupancy(lds_bytes, ctx->scene.general_materials);
  if (occupancy < 1) return 0;
  if (const char* env = getenv("YHAIR_ST_WAVES")) occupancy = std::max(1, std::min(occupancy, (atoi(env) + wpb - 1) / wpb));  // waves per CU
  const int64_t want = (pixels + (int64_t)P * wpb - 1) / ((int64_t)P * wpb);
  *slots_per_wave    = P;
  *grid_blocks       = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * occupancy));
  if (lds_out) *lds_out = lds_bytes;
  return 1;
}

// Hand-out order of the work items for the streaming integrator. Its waves take items four at a time (64 pixels)
// and keep them until all their samples are done, and the first R takes (R = what the path pools hold) are
// resident together: dealt from the cost-sorted list in order, the first waves would get all the expensive pixels
// and bound the launch (sparse hair: C1, C4). So each block of R takes is dealt like cards: take c holds one item
// of each quarter of the block, and consecutive takes are spread over the block by a golden-ratio stride — every
// wave gets a uniform sample of the costs, expensive blocks still come first.
static void deal_block(std::vector<int>& out, const int* items, size_t n, size_t R) {
  for (size_t b0 = 0; b0 < n; b0 += 4 * R) {
    const size_t M  = std::min(n - b0, 4 * R);
    const size_t Rb = (M + 3) / 4;  // takes in this block
    size_t       A  = std::max<size_t>(1, (size_t)(0.6180339887 * (double)Rb));
    auto gcd = [](size_t a, size_t b) { while (b) { size_t t = a % b; a = b, b = t; } return a; };
    while (gcd(A, Rb) != 1) A++;
    for (size_t c = 0; c < Rb; c++) {
      const size_t cp = (c * A) % Rb;
      for (size_t k = 0; k < 4; k++)
        if (cp + k * Rb < M) out.push_back(items[b0 + cp + k * Rb]);
    }
  }
}
static void deal_items_for_stream(yh_context* ctx, std::vector<int>& items) {
  int P = 0, grid = 0;
  ctx->state.num_groups = 1, ctx->state.group_begin[0] = 0, ctx->state.group_begin[1] = (int)items.size();
  if (items.empty() || !stream_geometry(ctx, (int)items.size(), &P, &grid, nullptr)) return;
  const size_t R = (size_t)grid * (yhk_stream_block_threads() / 64) * (size_t)(P / 64);  // takes resident together
  // Groups: one compact image region per XCD (yh_device.h: yhd_state::group_begin). The items in Morton order of
  // their tiles, cut into G runs of equal cost; inside a run the dealing above. MEASURED WITHOUT GAIN, so off by
  // default (G = 1; YHAIR_ST_GROUPS=8 turns it on): C3 296 -> 300 Msamples/s, C2 237 -> 222 (the regions' costs drift
  // apart during a launch), C4 unchanged (profiles/r02/k_stream_xcd_groups.txt) — after the first bounce the rays of
  // a region wander through the hair, and 4 MB of L2 hold little of a region's 40 MB anyway.
  int G = 1;
  if (const char* env = getenv("YHAIR_ST_GROUPS")) G = std::max(1, std::min(8, atoi(env)));
  if ((size_t)G * 64 > items.size()) G = 1;
  std::vector<int> out;
  out.reserve(items.size());
  if (G == 1) {
    deal_block(out, items.data(), items.size(), R);
  } else {
    auto morton = [&](int item) -> uint64_t {
      const int tile = item >> 2, tx = tile % ctx->state.tiles_x, ty = tile / ctx->state.tiles_x;
      const unsigned x = (unsigned)(2 * tx + (item & 1)), y = (unsigned)(2 * ty + ((item >> 1) & 1));  // 4x4-pixel quadrant coordinates
      uint64_t m = 0;
      for (int b = 0; b < 16; b++) m |= ((uint64_t)((x >> b) & 1) << (2 * b)) | ((uint64_t)((y >> b) & 1) << (2 * b + 1));
      return m;
    };
    std::vector<std::pair<uint64_t, int>> order;  // (morton, rank in the cost-sorted list)
    order.reserve(items.size());
    for (size_t i = 0; i < items.size(); i++) order.push_back({morton(items[i]), (int)i});
    std::sort(order.begin(), order.end());
    double total = 0;
    for (int it : items) total += 1.0 + (double)ctx->item_cost[(size_t)it];
    size_t at = 0;
    double acc = 0;
    for (int g = 0; g < G; g++) {
      std::vector<int> ranks;  // this group's items, by rank in the cost-sorted list (= most expensive first)
      const double upto = total * (g + 1) / G;
      while (at < order.size() && (g == G - 1 || acc < upto)) {
        acc += 1.0 + (double)ctx->item_cost[(size_t)items[(size_t)order[at].second]];
        ranks.push_back(order[at].second);
        at++;
      }
      std::sort(ranks.begin(), ranks.end());
      std::vector<int> grp;
      grp.reserve(ranks.size());
      for (int r : ranks) grp.push_back(items[(size_t)r]);
      ctx->state.group_begin[g] = (int)out.size();
      deal_block(out, grp.data(), grp.size(), std::max<size_t>(1, R / G));
    }
    ctx->state.num_groups = G, ctx->state.group_begin[G] = (int)out.size();
  }
  items.swap(out);
}

// One launch of the streaming integrator (csrc/stream.hip): persistent wavefronts, one path pool each, one lane per path.
static int stream_impl(yh_context* ctx, int nsamples, bool sync) {
  int P = 0, grid = 0, lds_bytes = 0;
  if (!stream_geometry(ctx, ctx->state.num_tiles, &P, &grid, &lds_bytes))
    return fail(ctx, YH_E_DEVICE, "k_stream cannot run with its LDS layout on this device");
  const int     wpb    = yhk_stream_block_threads() / 64;
  const size_t  waves  = (size_t)grid * wpb, slots = waves * P;
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
  if ((rc = ensure_lane_blob(ctx))) return rc;
  const bool prof = getenv("YHAIR_ST_PROF") && atoi(getenv("YHAIR_ST_PROF")) != 0;  // developer switch: per-stage counters on stderr
  if (prof) {
    if ((rc = alloc_zero(ctx, ctx->d_st_prof, 64 * 8))) return rc;
    ctx->stream_pool.prof = (unsigned long long*)ctx->d_st_prof.p;
  } else {
    ctx->stream_pool.prof = nullptr;
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
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->ev0, ctx->ev1));
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

static int trace_impl(yh_context* ctx, int nsamples, bool counted, bool sync) {
  if (!ctx) return YH_E_INVALID;
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
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (int rc = upload_work_items(ctx)) return rc;
    }
  }
  if (counted && (ctx->state.launch_shape == 5 || (ctx->state.launch_shape >= 4 && (ctx->scene.general_materials || ctx->state.launch_shape >= 7)))) {  // the octet kernel's list holds half-quadrant entries: the instrumented (quad) build needs its own
    ctx->launch_shape = ctx->state.launch_shape = 0;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (int rc = upload_work_items(ctx)) return rc;
  }
  int shape = path ? ctx->state.launch_shape : 0;  // the preview shaders have one launch shape
  if (counted && (shape == 3 || (shape >= 2 && ctx->scene.general_materials))) shape = shape == 3 ? 1 : 0;  // no instrumented build of k_stream, nor of the GENERAL 8-wide forms
  if (shape == 3 && !getenv("YHAIR_SHAPE")) {      // a candidate that cannot run here is dropped, not an error: k_trace renders the same bits
    int P = 0, grid = 0;
    if (!stream_geometry(ctx, ctx->state.num_tiles, &P, &grid, nullptr)) {
      ctx->shape_ms[3] = std::numeric_limits<double>::infinity();
      shape = ctx->dense > 0 ? 1 : 0;
      ctx->launch_shape = ctx->state.launch_shape = shape;
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (int rc = upload_work_items(ctx)) return rc;
    }
  }
  if ((shape == 4 || shape >= 6) && !counted && !getenv("YHAIR_SHAPE") &&
      yhk_trace_occupancy(yhk_trace_lds_bytes(&ctx->scene, shape), ctx->scene.general_materials, shape) < 1) {  // (likewise: a tree too deep for the wide forms' LDS stacks)
    ctx->shape_ms[shape] = std::numeric_limits<double>::infinity();
    shape = 0;
    ctx->launch_shape = ctx->state.launch_shape = shape;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (int rc = upload_work_items(ctx)) return rc;
  }
  if (shape == 2 || shape >= 4)
    if (int rc = ensure_wide_nodes(ctx)) return rc;
  ctx->last_shape = shape, ctx->last_counted = counted, ctx->planned_settled = ctx->costs_settled;
#ifdef YH_LAB_WAVEFRONT
  if (path && !counted && getenv("YHAIR_LAB_WAVEFRONT")) return wavefront_impl(ctx, nsamples, sync);  // developer build only (make WAVEFRONT=1)
#endif
  if (shape == 3) return stream_impl(ctx, nsamples, sync);
  if (shape == 2 && yhk_trace_occupancy(yhk_trace_lds_bytes(&ctx->scene, 2), ctx->scene.general_materials, 2) < 1)
    return fail(ctx, YH_E_INVALID, "launch shape 2 (quads over 8-wide nodes) is a developer kernel: build with make W8=1");
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
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->ev0, ctx->ev1));
    return replan_after_launch(ctx, nsamples);
  }
  return YH_OK;
}
// One side-by-side launch: k_trace_sbs over the whole list — its first G_o workgroups the octet entries behind the quad items,
// the other G_q the quad items [0, hy_quad_items).
static int side_by_side_impl(yh_context* ctx, int nsamples, bool sync) {
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
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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

int yh_download(yh_context* ctx, float* rgba) {
  if (!ctx || !rgba) return YH_E_INVALID;
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_download before yh_init_state");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t bytes = (size_t)ctx->state.width * ctx->state.height * 16;
  HIPCHK(ctx, hipMemsetAsync(ctx->d_image.p, 0, bytes, ctx->stream));
  int e = yhk_resolve(&ctx->state, (int)ctx->owned.size(), ctx->state.samples_done, ctx->d_image.p, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_resolve launch: %s", hipGetErrorString((hipError_t)e));
  HIPCHK(ctx, hipMemcpyAsync(rgba, ctx->d_image.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return YH_OK;
}

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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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

static void destroy_communicators(yh_context* ctx) {
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
      HIPCHK(root, hipStreamSynchronize(ctxs[i]->stream));
    }
  } else {
    for (int i = 0; i < n; i++) {  // device-to-device copies (contexts sharing a device, or YHAIR_GATHER=peer)
      yh_context* c = ctxs[i];
      HIPCHK(root, hipSetDevice(c->device));
      HIPCHK(root, hipStreamSynchronize(c->stream));
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
  HIPCHK(root, hipStreamSynchronize(root->stream));
  return YH_OK;
}

int yh_download_rng(yh_context* ctx, uint64_t* state_inc) {
  if (!ctx || !state_inc) return YH_E_INVALID;
  if (!ctx->have_state) return fail(ctx, YH_E_STATE, "yh_download_rng before yh_init_state");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t npix = (size_t)ctx->state.width * ctx->state.height;
  std::vector<uint64_t> st(npix), inc(npix);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(costs, ctx->d_tile_cost.p, (size_t)ctx->num_tiles_total * 16, hipMemcpyDeviceToHost));
  return YH_OK;
}

// ---- unit-level batches ----------------------------------------------------
namespace {
struct Staged {
  std::vector<DevBuf> bufs;
  yh_context*         ctx;
  int                 rc = YH_OK;
  explicit Staged(yh_context* c) : ctx(c) { bufs.reserve(8); }
  void* in(const void* src, size_t bytes) {
    bufs.emplace_back();
    if (rc == YH_OK) rc = upload(ctx, bufs.back(), src, bytes);
    return bufs.back().p;
  }
  void* out(size_t bytes) {
    bufs.emplace_back();
    if (rc == YH_OK) rc = alloc_zero(ctx, bufs.back(), bytes);
    return bufs.back().p;
  }
};
int finish(yh_context* ctx, int launch_err, void* dst, const void* src, size_t bytes) {
  if (launch_err) return fail(ctx, YH_E_DEVICE, "kernel launch: %s", hipGetErrorString((hipError_t)launch_err));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return YH_OK;
}
}  // namespace

int yh_hair_brdf_batch(yh_context* ctx, int n, const yh_material* materials, const float* v, const float* normal,
    const float* tangent, float* brdf) {
  if (!ctx || n < 0 || (n && (!materials || !v || !normal || !tangent || !brdf))) return YH_E_INVALID;
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   dm = s.in(materials, sizeof(yh_material) * (size_t)n);
  auto   dv = (float*)s.in(v, 4 * (size_t)n);
  auto   dn = (float*)s.in(normal, 12 * (size_t)n);
  auto   dt = (float*)s.in(tangent, 12 * (size_t)n);
  auto   o  = (float*)s.out(120 * (size_t)n);
  if (s.rc) return s.rc;
  return finish(ctx, yhk_hair_brdf(n, dm, dv, dn, dt, o, ctx->stream), brdf, o, 120 * (size_t)n);
}
static int wowi(yh_context* ctx, int n, const float* brdf, const float* a, size_t a_floats, const float* b,
    size_t b_floats, float* out, size_t out_floats, int which) {
  if (!ctx || n < 0 || (n && (!brdf || !a || !b || !out))) return YH_E_INVALID;
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   db = (float*)s.in(brdf, 120 * (size_t)n);
  auto   da = (float*)s.in(a, 4 * a_floats * n);
  auto   dbb = (float*)s.in(b, 4 * b_floats * n);
  auto   o  = (float*)s.out(4 * out_floats * n);
  if (s.rc) return s.rc;
  int e = which == 0   ? yhk_hair_eval(n, db, da, dbb, o, ctx->stream)
          : which == 1 ? yhk_hair_sample(n, db, da, dbb, o, ctx->stream)
                       : yhk_hair_pdf(n, db, da, dbb, o, ctx->stream);
  return finish(ctx, e, out, o, 4 * out_floats * n);
}
int yh_hair_eval_batch(yh_context* ctx, int n, const float* brdf, const float* wo, const float* wi, float* f) {
  return wowi(ctx, n, brdf, wo, 3, wi, 3, f, 3, 0);
}
int yh_hair_sample_batch(yh_context* ctx, int n, const float* brdf, const float* wo, const float* rn, float* wi) {
  return wowi(ctx, n, brdf, wo, 3, rn, 2, wi, 3, 1);
}
int yh_hair_pdf_batch(yh_context* ctx, int n, const float* brdf, const float* wo, const float* wi, float* pdf) {
  return wowi(ctx, n, brdf, wo, 3, wi, 3, pdf, 1, 2);
}
int yh_hair_eval_pdf_batch(yh_context* ctx, int n, const float* brdf, const float* wo, const float* wi, float* pdf) {
  return yh_hair_pdf_batch(ctx, n, brdf, wo, wi, pdf);
}

int yh_curves_to_lines(yh_context* ctx, int n, const float* P, const float* width0, const float* width1,
    int base_vertex, float* positions, float* normals, float* radius, int* lines) {
  if (!ctx || n < 0 || (n && (!P || !width0 || !width1 || !positions || !normals || !radius || !lines))) return YH_E_INVALID;
  if (n == 0) return YH_OK;
  if (n > 400000000 || base_vertex < 0 || (long long)base_vertex + 5ll * n > 2147483647ll)
    return fail(ctx, YH_E_INVALID, "too many curves for 32-bit vertex indices");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   dp = (float*)s.in(P, 48 * (size_t)n);
  auto   d0 = (float*)s.in(width0, 4 * (size_t)n);
  auto   d1 = (float*)s.in(width1, 4 * (size_t)n);
  auto   op = (float*)s.out(60 * (size_t)n);
  auto   on = (float*)s.out(60 * (size_t)n);
  auto   orad = (float*)s.out(20 * (size_t)n);
  auto   ol = (int*)s.out(32 * (size_t)n);
  if (s.rc) return s.rc;
  int e = yhk_curves_to_lines(n, dp, d0, d1, base_vertex, op, on, orad, ol, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "k_curves_to_lines launch: %s", hipGetErrorString((hipError_t)e));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(positions, op, 60 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(normals, on, 60 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(radius, orad, 20 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(lines, ol, 32 * (size_t)n, hipMemcpyDeviceToHost));
  return YH_OK;
}

// The same tree built on the device (csrc/bvh_gpu.hip): fills `tree` like yhh::build_bvh.
static int build_bvh_device(yh_context* ctx, const std::vector<yhh::Box>& boxes, yhh::Tree& tree) {
  int                n = (int)boxes.size(), num_nodes = 0, depth = 0;
  std::vector<float> nodes8((size_t)(2 * n + 1) * 8);
  tree.primitives.resize((size_t)n);
  static_assert(sizeof(yhh::Box) == 24, "boxes are passed as 6 floats");
  int e = yhk_bvh_build_gpu(n, (const float*)boxes.data(), nodes8.data(), tree.primitives.data(), &num_nodes, &depth, ctx->stream);
  if (e) return fail(ctx, YH_E_DEVICE, "device BVH build: %s", hipGetErrorString((hipError_t)e));
  tree.nodes.resize((size_t)num_nodes);
  tree.max_depth = depth;
  for (int i = 0; i < num_nodes; i++) {
    const float* o  = &nodes8[(size_t)i * 8];
    yhh::Node&   nd = tree.nodes[(size_t)i];
    memcpy(nd.bbox.min, o, 12), memcpy(nd.bbox.max, o + 3, 12);
    int start, meta;
    memcpy(&start, o + 6, 4), memcpy(&meta, o + 7, 4);
    nd.start = start, nd.num = (short)(meta & 0xFFFF), nd.internal = (meta >> 16) & 1, nd.axis = (unsigned char)((meta >> 24) & 3);
  }
  return YH_OK;
}
int yh_bvh_build_gpu(yh_context* ctx, int n, const float* boxes, float* nodes, int* primitives) {
  if (!ctx || n < 0 || (n && !boxes)) return YH_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<yhh::Box> b((size_t)n);
  if (n) memcpy((void*)b.data(), boxes, sizeof(yhh::Box) * (size_t)n);
  yhh::Tree tree;
  int       rc = build_bvh_device(ctx, b, tree);
  if (rc) return rc;
  if (nodes)
    for (size_t i = 0; i < tree.nodes.size(); i++) {
      auto&  nd = tree.nodes[i];
      float* o  = nodes + 8 * i;
      memcpy(o, nd.bbox.min, 12), memcpy(o + 3, nd.bbox.max, 12);
      int a = nd.start, c = (int)nd.num | ((int)nd.internal << 16) | ((int)nd.axis << 24);
      memcpy(o + 6, &a, 4), memcpy(o + 7, &c, 4);
    }
  if (primitives && n) memcpy(primitives, tree.primitives.data(), sizeof(int) * (size_t)n);
  return (int)tree.nodes.size();
}

int yh_bvh_build_wide(int n, const float* boxes, int width, float* slots) {
  if (n < 0 || (n && !boxes) || (width != 4 && width != 8 && width != 16)) return YH_E_INVALID;
  std::vector<yhh::Box> b((size_t)n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) b[(size_t)i].min[k] = boxes[6 * (size_t)i + k], b[(size_t)i].max[k] = boxes[6 * (size_t)i + 3 + k];
  yhh::Tree tree;
  yhh::build_bvh(tree, b);
  const void* data  = nullptr;
  size_t      count = 0;
  std::vector<yhh::WideNode>   w4;
  std::vector<yhh::WideNode8>  w8;
  std::vector<yhh::WideNode16> w16;
  if (width == 4) yhh::collapse_wide(tree, w4), data = w4.data(), count = w4.size();
  if (width == 8) yhh::collapse_wide8(tree, w8), data = w8.data(), count = w8.size();
  if (width == 16) yhh::collapse_wide16(tree, w16), data = w16.data(), count = w16.size();
  if (slots && count) memcpy(slots, data, count * (size_t)width * sizeof(yhh::WideSlot));
  return (int)count;
}

int yh_bvh_build(int n, const float* boxes, float* nodes, int* primitives) {
  if (n < 0 || (n && !boxes)) return YH_E_INVALID;
  std::vector<yhh::Box> b((size_t)n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) b[(size_t)i].min[k] = boxes[6 * (size_t)i + k], b[(size_t)i].max[k] = boxes[6 * (size_t)i + 3 + k];
  yhh::Tree tree;
  yhh::build_bvh(tree, b);
  if (nodes)
    for (size_t i = 0; i < tree.nodes.size(); i++) {
      auto&  nd = tree.nodes[i];
      float* o  = nodes + 8 * i;
      memcpy(o, nd.bbox.min, 12), memcpy(o + 3, nd.bbox.max, 12);
      int a = nd.start, c = (int)nd.num | ((int)nd.internal << 16) | ((int)nd.axis << 24);
      memcpy(o + 6, &a, 4), memcpy(o + 7, &c, 4);
    }
  if (primitives && n) memcpy(primitives, tree.primitives.data(), sizeof(int) * (size_t)n);
  return (int)tree.nodes.size();
}

int yh_surface_lobe_batch(yh_context* ctx, int kind, int n, const float* params, const float* normal,
    const float* outgoing, const float* incoming, const float* rn, float* out) {
  if (!ctx || n < 0 || kind < 0 || kind >= YH_LOBE_COUNT || (n && (!params || !normal || !outgoing || !incoming || !rn || !out)))
    return YH_E_INVALID;
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   dp = (float*)s.in(params, 32 * (size_t)n);
  auto   dn = (float*)s.in(normal, 12 * (size_t)n);
  auto   da = (float*)s.in(outgoing, 12 * (size_t)n);
  auto   db = (float*)s.in(incoming, 12 * (size_t)n);
  auto   dr = (float*)s.in(rn, 12 * (size_t)n);
  auto   o  = (float*)s.out(28 * (size_t)n);
  if (s.rc) return s.rc;
  return finish(ctx, yhk_surface_lobe(kind, n, dp, dn, da, db, dr, o, ctx->stream), out, o, 28 * (size_t)n);
}
int yh_surface_bsdf_batch(yh_context* ctx, int n, const yh_material* materials, const float* normal,
    const float* outgoing, const float* incoming, const float* rn, float* out) {
  if (!ctx || n < 0 || (n && (!materials || !normal || !outgoing || !incoming || !rn || !out))) return YH_E_INVALID;
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<yhd_material> mats((size_t)n);
  for (int i = 0; i < n; i++) make_material(materials[i], mats[(size_t)i]);
  Staged s(ctx);
  auto   dm = s.in(mats.data(), sizeof(yhd_material) * (size_t)n);
  auto   dn = (float*)s.in(normal, 12 * (size_t)n);
  auto   da = (float*)s.in(outgoing, 12 * (size_t)n);
  auto   db = (float*)s.in(incoming, 12 * (size_t)n);
  auto   dr = (float*)s.in(rn, 12 * (size_t)n);
  auto   o  = (float*)s.out(4 * YH_SURFACE_BSDF_FLOATS * (size_t)n);
  if (s.rc) return s.rc;
  return finish(ctx, yhk_surface_bsdf(n, dm, dn, da, db, dr, o, ctx->stream), out, o, 4 * YH_SURFACE_BSDF_FLOATS * (size_t)n);
}

int yh_intersect_batch(yh_context* ctx, int n, const float* rays, int* object, int* element, float* uv,
    float* distance) {
  if (!ctx || n < 0 || (n && (!rays || !object || !element || !uv || !distance))) return YH_E_INVALID;
  if (!ctx->have_scene) return fail(ctx, YH_E_STATE, "yh_intersect_batch before yh_upload_scene");
  if (n == 0) return YH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staged s(ctx);
  auto   dr = (float*)s.in(rays, 32 * (size_t)n);
  auto   dob = (int*)s.out(4 * (size_t)n);
  auto   del = (int*)s.out(4 * (size_t)n);
  auto   duv = (float*)s.out(8 * (size_t)n);
  auto   dd  = (float*)s.out(4 * (size_t)n);
  if (s.rc) return s.rc;
  // Large batches of rays that start at the reference's ray_eps (every ray the path tracer itself makes) go one lane
  // per ray through the trace-only kernel (csrc/stream.hip: k_intersect_lanes), five waves per SIMD; small
  // ones, and rays with another tmin, a quad per ray (k_intersect). Same closest hits either way.
  // YHAIR_INTERSECT=quad | lane4 | lane5 | lane6 | lane8: developer switch (waves per SIMD of the lane kernel).
  const char* mode  = getenv("YHAIR_INTERSECT");
  int         waves = 5;  // 91 registers without a spill: five waves per SIMD (6 and 8 spill 39 / 61 registers, measured slower)
  bool        lanes = n >= 65536;
  if (mode && !strcmp(mode, "quad")) lanes = false;
  else if (mode && !strncmp(mode, "lane", 4)) lanes = true, waves = std::max(4, std::min(8, atoi(mode + 4)));
  for (int i = 0; lanes && i < n; i++) lanes = rays[8 * (size_t)i + 6] == 1e-4f;
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  if (lanes) {
    const int occupancy = std::min(waves, yhk_intersect_lanes_occupancy(&ctx->scene, waves));  // (256-thread blocks: one wave per SIMD each)
    if (occupancy < 1) return fail(ctx, YH_E_DEVICE, "k_intersect_lanes cannot run with its LDS layout on this device");
    const int    grid        = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)n + 255) / 256, (int64_t)ctx->num_cus * occupancy));
    const int    ovf_entries = 2 * std::max(8, ctx->stack_need);
    auto         dcur        = (int*)s.out(16);
    auto         dovf        = (unsigned int*)s.out((size_t)grid * 4 * ovf_entries * 64 * 4);
    if (s.rc) return s.rc;
    if (int rcb = ensure_lane_blob(ctx)) return rcb;
    if (!ctx->d_scene_copy.p) {
      int rc = upload(ctx, ctx->d_scene_copy, &ctx->scene, sizeof(yhd_scene));
      if (rc) return rc;
    }
    int e = yhk_intersect_lanes(&ctx->scene, (const yhd_scene*)ctx->d_scene_copy.p, n, dr, dcur, dovf, ovf_entries, dob, del, duv, dd, waves, grid, ctx->stream);
    if (e) return fail(ctx, YH_E_DEVICE, "k_intersect_lanes launch: %s", hipGetErrorString((hipError_t)e));
  } else {
    int e = yhk_intersect(&ctx->scene, n, dr, dob, del, duv, dd, ctx->stream);
    if (e) return fail(ctx, YH_E_DEVICE, "k_intersect launch: %s", hipGetErrorString((hipError_t)e));
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipEventElapsedTime(&ctx->last_ms, ctx->ev0, ctx->ev1));  // (yh_last_trace_ms: the kernel alone, without the copies)
  ctx->last_launches = 1;
  HIPCHK(ctx, hipMemcpy(object, dob, 4 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(element, del, 4 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(uv, duv, 8 * (size_t)n, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(distance, dd, 4 * (size_t)n, hipMemcpyDeviceToHost));
  return YH_OK;
}

// ---- the four self-tests (ext.cpp:555-693) ---------------------------------
// The host replays the reference's serial structure (seed, loop bounds with
// the accumulating float counters, per-block draw counts) and hands every
// (beta_m, beta_n) block to the device with the generator state at its start.
int yh_selftest(yh_context* ctx, int which, float* worst) {
  if (!ctx) return YH_E_INVALID;
  if (which < 0 || which > 3) return fail(ctx, YH_E_INVALID, "unknown self-test %d", which);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf sums, wbits;
  int    rc;
  if ((rc = alloc_zero(ctx, sums, 6 * sizeof(double)))) return rc;
  if ((rc = alloc_zero(ctx, wbits, 4))) return rc;
  auto lum = [](const double* s) { return (float)(0.2126 * s[0] + 0.7152 * s[1] + 0.0722 * s[2]); };
  auto sample_sphere = [](float rx, float ry, float* w) {  // math.h:4847-4852
    float z = 2 * ry - 1;
    float r = std::sqrt(fmin_(fmax_(1 - z * z, 0.0f), 1.0f));
    float phi = 2 * pif * rx;
    w[0] = r * std::cos(phi), w[1] = r * std::sin(phi), w[2] = z;
  };
  Rng   rng = make_rng(199382389514ULL);
  float wo[3] = {0, 0, 1};
  if (which == 0 || which == 1) {
    float x = rand1f(rng), y = rand1f(rng);
    sample_sphere(x, y, wo);
  }
  bool  ok  = true;
  float dev = 0;
  auto run = [&](float bm, float bn, int count, int per_iter, double* out, float* dmax) -> int {
    HIPCHK(ctx, hipMemsetAsync(sums.p, 0, 6 * sizeof(double), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(wbits.p, 0, 4, ctx->stream));
    int e = yhk_selftest(which, bm, bn, rng.state, rng.inc, count, wo, (double*)sums.p, (unsigned int*)wbits.p, ctx->stream);
    if (e) return fail(ctx, YH_E_DEVICE, "k_selftest launch: %s", hipGetErrorString((hipError_t)e));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out, sums.p, 6 * sizeof(double), hipMemcpyDeviceToHost));
    unsigned int bits;
    HIPCHK(ctx, hipMemcpy(&bits, wbits.p, 4, hipMemcpyDeviceToHost));
    memcpy(dmax, &bits, 4);
    skip_rng(rng, (uint64_t)count * per_iter);
    return YH_OK;
  };
  double s[6];
  float  d;
  if (which == 0 || which == 1) {
    for (float bm = 0.1f; bm < 1.0f; bm += 0.2f)
      for (float bn = 0.1f; bn < 1.0f; bn += 0.2f) {
        const int count = 300000;
        if ((rc = run(bm, bn, count, 3, s, &d))) return rc;
        float avg = which == 0 ? lum(s) / (count * (1 / (4 * pif))) : lum(s) / count;
        float lo = which == 0 ? 0.95f : 0.99f, hi = which == 0 ? 1.05f : 1.01f;
        if (!(avg >= lo && avg <= hi)) ok = false;
        dev = fmax_(dev, std::fabs(avg - 1));
      }
  } else if (which == 2) {
    for (float bm = 0.1f; bm < 1.0f; bm += 0.2f)
      for (float bn = 0.4f; bn < 1.0f; bn += 0.2f) {
        if ((rc = run(bm, bn, 10000, 5, s, &d))) return rc;
        if (!(d <= 0.001f)) ok = false;
        dev = fmax_(dev, d);
      }
  } else {
    for (float bm = 0.2f; bm < 1.0f; bm += 0.2f)
      for (float bn = 0.4f; bn < 1.0f; bn += 0.2f) {
        const int count = 64 * 1024;
        float x = rand1f(rng), y = rand1f(rng);
        sample_sphere(x, y, wo);
        if ((rc = run(bm, bn, count, 3, s, &d))) return rc;
        float fi = lum(s) / count, fu = lum(s + 3) / (count * (1 / (4 * pif)));
        float err = std::fabs(fi - fu) / fu;
        if (err >= 0.05f) ok = false;
        dev = fmax_(dev, err);
      }
  }
  if (worst) *worst = dev;
  if (!ok) return fail(ctx, YH_E_SELFTEST, "TEST FAILED! (self-test %d, worst deviation %g)", which, dev);
  return YH_OK;
}

}  // extern "C"
