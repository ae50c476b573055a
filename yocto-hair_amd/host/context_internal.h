// context_internal.h — what the translation units of the host library share: the launchers of csrc/*.hip, the context, small
// host helpers with the reference's operation order, and the internal functions that cross files. Not installed; the public
// interface is include/yhair.h.
//   context.cpp        create / destroy, errors, shard, downloads
//   scene_upload.cpp   yh_upload_scene = init_bvh + init_lights (pt.cpp:755-818,1695-1740): reference-identical BVHs, leaf-ordered
//                      records, inverse frames, per-material hair constants, light CDFs; the wide-node and lane-blob arrays
//   launch_plan.cpp    which kernel runs (timing trials, their record in memory and on disk) and the hand-out order of the work items
//   trace_launch.cpp   yh_init_state (pt.cpp:1931-1946) and the launches: yh_trace_samples and friends
//   gather.cpp         tile packing and the one collective (yh_gather_framebuffer: RCCL or peer copies)
//   batch_api.cpp      the unit-level batch entry points (hair BSDF, intersection, BVH build, curves, self-tests)
#ifndef YH_CONTEXT_INTERNAL_H_
#define YH_CONTEXT_INTERNAL_H_
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
#include "deadline.h"
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
int yhk_stream(const yhd_scene*, const yhd_scene* sc_dev, const yhd_state*, int, const yhd_stream*, int grid_blocks, hipStream_t);
int yhk_lane_tests(const yhd_float4* prims, yhd_float4* blob, int kind, int prim_base, int num_prims, long long test_off, hipStream_t);
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
// csrc/bvh_gpu.hip, device in / device out (yh_upload_scene): bounds, the tree, the leaf records and the wide collapses of one shape
int yhk_prim_boxes(int lines, int n, const float* pos, const float* radius, const int* idx, float* boxes, hipStream_t);
int yhk_bvh_build_resident(int n, const float* d_boxes, float* d_nodes, int* d_pid, int* num_nodes, int* levels, int* level_first, hipStream_t);
int yhk_leaf_records(int lines, int n, const int* pid, const float* pos, const float* nrm, const float* radius, const int* idx, void* prims_out, hipStream_t);
int yhk_wide_index(int num_nodes, const float* d_nodes, int levels, const int* level_first, int L, unsigned int* d_flag, unsigned int* d_widx, int* count, hipStream_t);
int yhk_wide_collapse(int L, int num_nodes, const float* d_nodes, const unsigned int* d_flag, const unsigned int* d_widx, int lines, long long node_off, long long test_off, void* blob, hipStream_t);
int yhk_curves_to_lines(int, const float*, const float*, const float*, int, float*, float*, float*, int*, hipStream_t);
int yhk_surface_lobe(int, int, const float*, const float*, const float*, const float*, const float*, float*, hipStream_t);
int yhk_surface_bsdf(int, const void*, const float*, const float*, const float*, const float*, float*, hipStream_t);
int yhk_selftest(int, float, float, uint64_t, uint64_t, int, const float*, double*, unsigned int*, hipStream_t);
}

// A device allocation owned by the context.
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

namespace {



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

constexpr int YH_SHAPES = 9;  // launch shapes: 0, 1 k_trace (4-wide nodes) | 2 (was: quads over 8-wide nodes; not built) | 3 k_stream | 4 k_trace with octets | 5 quads and octets side by side | 6 k_trace with sixteen lanes per path | 7 octets with leaf pairs | 8 sixteen lanes with leaf groups
struct yh_context {
  int         device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t  ev0 = nullptr, ev1 = nullptr;
  int         hy_quad_items = 0, hy_oct_entries = 0;  // layout of the work list for shape 5: [quad items][octet entries]
  std::vector<int> hy_oct_items;                       // ... and the items that run as octets
  int         num_cus = 0;
  std::string device_name;  // gcnArchName / marketing name / CU count: part of the key of the trial record on disk
  std::string error = "no error";
  yhh::BoundedCall sync_call;    // the blocking hipStreamSynchronize of this context's stream, on a worker thread with a deadline (wait_for_launch)
  bool        poisoned = false;  // a launch of this context did not complete within its deadline (wait_for_launch): no further launches, nothing is freed
  // scene
  bool      have_scene = false;
  yhd_scene scene{};
  DevBuf    d_prims, d_vpos, d_elems, d_objects, d_materials, d_scene_nodes,
      d_scene_prims, d_light_cdf, d_env_texels, d_light_table, d_env_tab;
  int       stack_need = 0, stack_need8 = 0, stack_need16 = 0;
  // where every shape's test records and nodes sit in the traversal kernels' array (yhd_scene::lane_blob), made at yh_upload_scene
  struct LaneShape { int kind, num_nodes, prim_base, num_prims; long long node_off, test_off, node_off8, node_off16; };  // offsets in 32-byte units: the shape's 4- / 8- / 16-wide nodes and its test records
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
  int              last_nsamples = 0;   // samples of the launch the item costs come from
  unsigned         launches_of_image = 0;  // synchronous launches since this IMAGE (scene, resolution, sampler, bounces) was first initialised: the re-planning schedule. A
                                           // re-initialised render of a known image is planned already (item_cost survives it) and goes on where the schedule was
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
  bool             trials_on_disk = false;          // ... or has been written there by this context (once per image)
  bool             have_costs = false;
  bool             costs_settled = false;   // the item costs come from a launch of at least YH_TRIAL_SPP samples (not from the 1-spp probe)
  bool             planned_settled = false; // ... and the most recent launch was planned from such costs (only then does its time rank a kernel)
  int              dense = -1;
  int              chain16 = -1; // 1: ... and four times as many: the sixteen-lane form (shape 6) is a candidate too
  int              chain = -1;   // 1: so few expensive items that even twice as many waves would all be resident: the launch is bound by the
                                 // chain of steps of ONE path, and the octet kernel (half the paths per wave, shape 4) is a candidate
  // path pool of the streaming integrator (csrc/stream.hip): per-wave slots, allocated at its first launch
  DevBuf           d_st_slots, d_st_medium, d_st_ovf, d_st_prof, d_st_wave_log, d_st_wave_begin, d_st_wave_fill, d_scene_copy;
  int              st_items = 0;         // work items of the list k_stream's hand-out was made for (deal_items_for_stream): what its launch geometry follows
  size_t           st_share_waves = 0;   // waves the per-wave shares of the work list were made for (0: none, everything through the cursor)
  int              st_share_slots = 0;   // ... and the pool slots per wave (a share holds at most slots / 16 items)
  std::vector<int>    st_share_begin, st_share_items;  // host copy of the shares in effect: offsets per wave, items in list order
  std::vector<double> st_share_cost;                   // ... and the cost each item was planned with
  std::vector<unsigned long long> st_last_log;         // the last k_stream launch's stamps per wave {begin, end}
  bool                st_log_fresh = false;            // ... not yet used by a hand-out, and taken on the shares above
  std::vector<float>  st_wave_speed;                   // per wave of the shares' launch geometry: its speed relative to its dispatch round's (deal_shares_by_speed)
  std::vector<float>  item_scale;                      // per work item: correction of its reported cost (BVH steps) towards the time it takes (deal_shares_by_speed)
  std::vector<double> stream_speed = {1.10, 1.045, 0.97, 0.885};  // relative speed of k_stream's waves by dispatch round = hardware wave slot (prior: C2's log; every launch refines it)
  size_t           st_slots = 0, st_medium_slots = 0, st_ovf_words = 0;
  yhd_stream       stream_pool{};
};

#pragma GCC visibility push(hidden)  // internal to libyhair.so
extern std::string g_create_error;  // why yh_create returned NULL
#define HIPCHK(ctx, call)                                                                               \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return fail(ctx, YH_E_DEVICE, "%s: %s", #call, hipGetErrorString(e_));       \
  } while (0)

// every wait for the context's stream is bounded (wait_for_launch, trace_launch.cpp)
#define YH_WAIT(ctx)                                      \
  do {                                                    \
    if (int wrc_ = wait_for_launch(ctx)) return wrc_;     \
  } while (0)

constexpr int YH_TRIAL_SPP = 32;  // shorter launches have flat, noisy costs: they neither rank kernels nor try new ones
constexpr double YH_TRIAL_TIE  = 1.15;
constexpr double YH_FINAL_TIE  = 1.05;  // after the trials: candidates this close to the fastest count as tied (pick_launch_shape)
constexpr int    YH_TRIALS_MAX = 2;

// ---- internal functions that cross translation units (defined in the file the comment above names) ----
int fail(yh_context* ctx, int code, const char* fmt, ...);
int upload(yh_context* ctx, DevBuf& buf, const void* src, size_t bytes);
int upload_keep(yh_context* ctx, DevBuf& buf, const void* src, size_t bytes);  // ... into a buffer that is kept while it is large enough
int alloc_zero(yh_context* ctx, DevBuf& buf, size_t bytes);
yhd_float4 node_lo(const yhh::Node& n);
yhd_float4 node_hi(const yhh::Node& n);
int choose_launch_shape(const yh_context* ctx);
void trials_load(yh_context* ctx);
bool trials_off();
void record_launch(yh_context* ctx, int nsamples, bool fresh_costs);
bool trial_pending(const yh_context* ctx);
int pick_launch_shape(const yh_context* ctx, int nsamples);
void build_work_items(const yh_context* ctx, std::vector<int>& items);
void split_items_for_octets(std::vector<int>& items);
void split_items_for_hex(std::vector<int>& items);
int upload_work_items(yh_context* ctx);
int expensive_items(const yh_context* ctx, const std::vector<int>& items);
bool side_by_side_grids(const yh_context* ctx, int* oct_blocks, int* quad_blocks);
void split_items_side_by_side(yh_context* ctx, std::vector<int>& items);
void lay_out_range(const yh_context* ctx, int* items, size_t n, int wpb, int G, int block_offset = 0);
void lay_out_first_round(const yh_context* ctx, std::vector<int>& items, int shape);
int replan_after_launch(yh_context* ctx, int nsamples);
bool lane_kernels_can_address(const yh_context* ctx);  // launch_plan.cpp: the lane blob fits the one-lane kernels' 32-bit offsets
int stream_geometry(const yh_context* ctx, int num_items, int* slots_per_wave, int* grid_blocks, int* lds_out, bool* single_generation = nullptr);
void note_stream_wave_log(yh_context* ctx, const unsigned long long* log, size_t waves);
void deal_items_for_stream(yh_context* ctx, std::vector<int>& items);
int build_bvh_device(yh_context* ctx, const std::vector<yhh::Box>& boxes, yhh::Tree& tree);
int stream_impl(yh_context* ctx, int nsamples, bool sync);
int trace_impl(yh_context* ctx, int nsamples, bool counted, bool sync);
int wait_for_launch(yh_context* ctx);  // hipStreamSynchronize(ctx->stream) with a deadline (host/deadline.h)
int side_by_side_impl(yh_context* ctx, int nsamples, bool sync);
void destroy_communicators(yh_context* ctx);
inline int tiles_of(int n) { return (n + YH_TILE - 1) / YH_TILE; }
void make_material(const yh_material& m, yhd_material& d);  // scene_upload.cpp
#pragma GCC visibility pop
#endif
