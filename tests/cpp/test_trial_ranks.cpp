// CPU-side rehearsal of EIGHT COLD RANKS ON ONE NODE: the kernel-trial bookkeeping of host/launch_plan.cpp — the REAL file,
// compiled into this test — driven by eight "ranks" (threads, one yh_context each, shard r of 8) against ONE trials_v2.txt, with the device
// replaced by a time model. Compiled and run by tests/test_abi.py::test_eight_cold_ranks_share_one_trial_record (g++ only, no HIP runtime, no GPU).
//
//   test_trial_ranks cold DIR    eight ranks start together on an empty record: every rank runs its own trial sequence, writes ONE complete line
//                                (concurrent O_APPEND writers), ends with a complete record, and none of its timed steps contains a trial;
//                                ranks whose candidates tie within YH_FINAL_TIE settle on the SAME kernel; then 8 x 40 more images for the file.
//   test_trial_ranks warm DIR R:S ...   a second process on the same directory (the Python side has meanwhile appended a torn line, a damaged
//                                line and a second record for one key): no rank runs a trial, every first request is ONE launch, nobody writes;
//                                R:S = rank R must run shape S (the LAST complete line of a key counts, torn and damaged ones do not).
//
// What is mocked: the launch itself (trace_impl's bookkeeping restated in fake_launch below, the kernel's duration and item costs from a model)
// and the five launch-geometry queries of csrc/*.hip. What is real: pick_launch_shape, trial_pending, record_launch, trials_load / trials_store,
// the record's line format, its O_APPEND writes and its reader, replan_after_launch, upload_work_items and the hand-out layouts.
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <fstream>
#include <set>
#include <sstream>

#include "../../yocto-hair_amd/host/launch_plan.cpp"

// ---- the device side, mocked ------------------------------------------------------------------------------------------------------------
extern "C" {
hipError_t  hipFree(void* p) { free(p); return hipSuccess; }
hipError_t  hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) { memcpy(dst, src, n); return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "mock"; }
int yhk_block_threads(int shape) { return shape == 0 || shape == 5 ? 512 : 256; }
int yhk_trace_lds_bytes(const yhd_scene*, int) { return 32768; }
int yhk_trace_occupancy(int, int, int shape) { return yhk_block_threads(shape) == 512 ? 2 : 4; }  // 16 waves per CU either way
int yhk_trace_sbs_lds_bytes(const yhd_scene*) { return 32768; }
int yhk_trace_sbs_occupancy(int, int) { return 2; }
int yhk_stream_block_threads(void) { return 256; }
int yhk_stream_lds_bytes(int, int slots) { return 16384 + 4 * (64 * 16 * 4 + 512 + 12 * slots); }
int yhk_stream_occupancy(int, int) { return 4; }
}
int fail(yh_context* ctx, int code, const char* fmt, ...) {
  char    buf[256];
  va_list ap;
  va_start(ap, fmt), vsnprintf(buf, sizeof(buf), fmt, ap), va_end(ap);
  if (ctx) ctx->error = buf;
  return code;
}
int upload_keep(yh_context*, DevBuf& buf, const void* src, size_t bytes) {
  if (buf.bytes < bytes) buf.reset(), buf.p = malloc(bytes), buf.bytes = bytes;
  memcpy(buf.p, src, bytes);
  return YH_OK;
}

static std::atomic<int> fails{0};
#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) printf("FAILED line %d: %s\n", __LINE__, #cond), fails++;     \
  } while (0)

// ---- one rank ----------------------------------------------------------------------------------------------------------------------------
struct Lcg {
  uint64_t s;
  double   next() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) / 9007199254740992.0; }
};
// ms per sample of each launch shape on this rank's shard. CHAIN: a shard of a sparse image (C1 on one of eight GPUs) — the wider the form the
// shorter the chain, the two sixteen-lane forms within 3 % of each other (a tie: the fixed order has to decide it the same way on every rank,
// whichever of the two a rank's own noise puts first). DENSE: every item expensive, k_stream ahead.
static double model_ms(bool dense, int shape, int rank) {
  static const double chain_ms[YH_SHAPES] = {0.196, 0.230, 0, 0.9, 0.153, 0.21, 0.128, 0.150, 0.1255};
  static const double dense_ms[YH_SHAPES] = {2.9, 2.6, 0, 2.1, 4.0, 3.0, 5.0, 4.2, 5.2};
  const double rank_factor = 1.0 + 0.004 * ((rank * 5) % 8 - 3.5);  // the GPUs of a node are not equally fast
  return (dense ? dense_ms : chain_ms)[shape] * rank_factor;
}
struct Rank {
  yh_context ctx{};
  bool       dense = false;
  Lcg        rng{1};
  std::vector<unsigned int> base_cost;  // per work item: its cost per sample
  int        launches = 0, trials_seen = 0;

  void init(int rank, int world, int w, int h, bool dense_, uint64_t scene_key) {
    dense = dense_, rng.s = 977 * (uint64_t)(rank + 1) + (uint64_t)w;
    ctx.device_name = "gfx950/mock/256", ctx.num_cus = 256, ctx.rank = rank, ctx.world = world, ctx.scene_key = scene_key;
    ctx.have_scene = true;
    // yh_init_state's bookkeeping (trace_launch.cpp) for a NEW image
    ctx.params.resolution = w, ctx.params.bounces = 8, ctx.params.shader = YH_SHADER_PATH;
    const int tx = tiles_of(w), ty = tiles_of(h);
    ctx.num_tiles_total = tx * ty;
    ctx.owned.clear();
    for (int t = rank; t < ctx.num_tiles_total; t += world) ctx.owned.push_back(t);
    ctx.item_cost.assign((size_t)ctx.num_tiles_total * 4, 0);
    ctx.have_costs = false, ctx.costs_settled = false, ctx.dense = -1, ctx.chain = -1, ctx.chain16 = -1, ctx.launch_shape = 0;
    ctx.state.width = w, ctx.state.height = h, ctx.state.tiles_x = tx, ctx.state.bounces = 8, ctx.state.shader = YH_SHADER_PATH, ctx.state.launch_shape = 0;
    ctx.state.shard_rank = rank, ctx.state.shard_world = world;
    ctx.d_tile_cost.p = calloc((size_t)ctx.num_tiles_total * 4, 4), ctx.d_tile_cost.bytes = (size_t)ctx.num_tiles_total * 16;
    ctx.d_tiles.p = calloc((size_t)ctx.num_tiles_total * 16 + 16, 4), ctx.d_tiles.bytes = ((size_t)ctx.num_tiles_total * 16 + 16) * 4;
    base_cost.assign(ctx.item_cost.size(), 0);
    for (int t : ctx.owned)
      for (int q = 0; q < 4; q++) base_cost[(size_t)t * 4 + q] = dense ? 900 + (unsigned)(rng.next() * 200) : (rng.next() < 0.1 ? 900 + (unsigned)(rng.next() * 200) : 8);
    ctx.have_state = true, ctx.launches_of_image = 0;
    trials_load(&ctx);
    // the 1-spp probe of a new image (yh_init_state): costs, no ranking
    fake_launch(1);
    ctx.state.samples_done = 0, ctx.launches_of_image = 0, ctx.last_shape = -1, ctx.last_ms = 0;
    ctx.have_costs = true;
    launches = 0, trials_seen = 0;
  }
  // trace_impl (trace_launch.cpp): the kernel for this launch, the list for it, the launch, the bookkeeping behind it
  int fake_launch(int n) {
    const int want = pick_launch_shape(&ctx, n);
    if (want != ctx.state.launch_shape) {
      ctx.launch_shape = ctx.state.launch_shape = want;
      if (int rc = upload_work_items(&ctx)) return rc;
    }
    const int shape = ctx.state.launch_shape;
    ctx.last_shape = shape, ctx.last_counted = false, ctx.planned_settled = ctx.costs_settled;
    const bool was_trial = ctx.costs_settled && n >= YH_TRIAL_SPP && n < 2 * YH_TRIAL_SPP;
    ctx.last_ms = (float)(model_ms(dense, shape, ctx.rank) * n * (1.0 + 0.012 * (rng.next() - 0.5)) + 0.26);  // (0.26 ms fixed per launch)
    unsigned int* cost = (unsigned int*)ctx.d_tile_cost.p;
    for (int t : ctx.owned)
      for (int q = 0; q < 4; q++) cost[(size_t)t * 4 + q] = (unsigned)(base_cost[(size_t)t * 4 + q] * (double)n * (0.9 + 0.2 * rng.next()));
    ctx.state.samples_done += n, ctx.last_launches = 1;
    launches++, trials_seen += was_trial ? 1 : 0;
    return replan_after_launch(&ctx, n);
  }
  // yh_trace_samples (trace_launch.cpp): a long request starts with the short trial launches of the kernels this image has not timed yet
  int trace_samples(int nsamples) {
    int remaining = nsamples;
    do {
      const int n = (remaining >= 2 * YH_TRIAL_SPP && trial_pending(&ctx)) ? YH_TRIAL_SPP : remaining;
      if (int rc = fake_launch(n)) return rc;
      remaining -= n;
    } while (remaining > 0);
    return YH_OK;
  }
  // bench.py's timed_run: W warm-up steps, then requests of 64 until no trial is pending, then K timed steps. Returns the launches inside the timed steps.
  int bench(int warmup, int steps, int spp, int* shape_out, bool* one_kernel) {
    for (int k = 0; k < warmup; k++) trace_samples(spp);
    for (int extra = 0; trial_pending(&ctx) && extra < 16; extra++) trace_samples(64);
    CHECK(!trial_pending(&ctx));
    launches = 0;
    std::set<int> shapes;
    for (int k = 0; k < steps; k++) trace_samples(spp), shapes.insert(ctx.last_shape);
    *shape_out = ctx.last_shape, *one_kernel = shapes.size() == 1;
    return launches;
  }
};

static std::vector<std::string> read_lines(const std::string& path) {
  std::vector<std::string> v;
  std::ifstream f(path);
  for (std::string l; std::getline(f, l);) v.push_back(l);
  return v;
}
// a record line as disk_store writes it: key, " =", YH_SHAPES pairs "ms:trials", " ; dense chain chain16"
static bool line_is_complete(const std::string& l) {
  const size_t eq = l.find(" =");
  if (eq == std::string::npos) return false;
  const char* p = l.c_str() + eq + 2;
  for (int k = 0; k < YH_SHAPES; k++) {
    double ms;
    int    tr, n = 0;
    if (sscanf(p, " %lf:%d%n", &ms, &tr, &n) != 2) return false;
    p += n;
  }
  int d, c, c16, n = 0;
  return sscanf(p, " ; %d %d %d%n", &d, &c, &c16, &n) == 3 && p[n] == 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1], dir = argv[2], file = dir + "/trials_v2.txt";
  setenv("YHAIR_CACHE_DIR", dir.c_str(), 1);
  unsetenv("YHAIR_NO_DISK_CACHE"), unsetenv("YHAIR_NO_TRIAL_CACHE"), unsetenv("YHAIR_SHAPE"), unsetenv("YHAIR_NO_TRIALS"), unsetenv("YHAIR_DEVICE_SHARE");
  constexpr int R = 8;
  // all ranks leave the gate together: the trial sequences and the record's writes overlap as they would on a node
  std::mutex              gm;
  std::condition_variable gcv;
  int                     waiting = 0, generation = 0;
  auto gate = [&] {
    std::unique_lock<std::mutex> lock(gm);
    const int g = generation;
    if (++waiting == R) waiting = 0, generation++, gcv.notify_all();
    else gcv.wait(lock, [&] { return generation != g; });
  };
  int  shape[2][R], timed_launches[2][R];
  bool one_kernel[2][R];
  std::vector<std::thread> th;
  if (mode == "cold") {
    CHECK(read_lines(file).empty());
    for (int r = 0; r < R; r++)
      th.emplace_back([&, r] {
        for (int sc = 0; sc < 2; sc++) {  // sc 0: C1 720^2 sharded eight ways (chain-bound: five candidates); sc 1: C3 1280^2 (dense: three)
          Rank k;
          gate();
          k.init(r, R, sc ? 1280 : 720, sc ? 1280 : 720, sc == 1, sc ? 0xC3 : 0xC1);
          CHECK(k.ctx.trials_from_disk == false);
          CHECK(trial_pending(&k.ctx));  // a cold rank has everything to try
          timed_launches[sc][r] = k.bench(5, 20, sc ? 512 : 77, &shape[sc][r], &one_kernel[sc][r]);
          CHECK(k.trials_seen >= (sc ? 3 : 5));  // every candidate was tried at least once
          int cand[6], n = candidates(&k.ctx, cand);
          CHECK(n == (sc ? 3 : 5));
          for (int c = 0; c < n; c++) CHECK(k.ctx.shape_ms[cand[c]] > 0 && k.ctx.shape_trials[cand[c]] >= 1);  // a complete record
          CHECK(k.ctx.trials_on_disk);
          CHECK(k.ctx.dense == (sc ? 1 : 0) && k.ctx.chain == (sc ? 0 : 1));
        }
        // ... and forty more images per rank, for the file: 8 x 40 concurrent single-line appends
        for (int img = 0; img < 40; img++) {
          Rank k;
          k.init(r, R, 64 + 8 * img, 64 + 8 * img, img % 2 == 1, 0xAB);
          int  s;
          bool one;
          (void)k.bench(1, 2, 77, &s, &one);
          CHECK(k.ctx.trials_on_disk);
        }
      });
    for (auto& t : th) t.join();
    for (int sc = 0; sc < 2; sc++)
      for (int r = 0; r < R; r++) {
        CHECK(timed_launches[sc][r] == 20);  // launches_in_timed_steps == steps: no trial inside any rank's timed region
        CHECK(one_kernel[sc][r]);
        CHECK(shape[sc][r] == shape[sc][0]);  // ranks whose candidates tie settle on ONE kernel (the fixed order decides, not a rank's noise)
      }
    CHECK(shape[0][0] == 8 && shape[1][0] == 3);
    const auto lines = read_lines(file);
    CHECK(lines.size() == (size_t)R * 42);  // one line per (rank, image): nobody wrote twice, nobody's line was lost
    std::set<std::string> keys;
    for (const auto& l : lines) {
      CHECK(line_is_complete(l));
      keys.insert(l.substr(0, l.find(" =")));
    }
    CHECK(keys.size() == lines.size());  // no torn or interleaved line: every key is whole and distinct
    printf("cold: %zu lines, chain-bound shard -> shape %d on all ranks, dense -> shape %d\n", lines.size(), shape[0][0], shape[1][0]);
  } else {
    const size_t before = read_lines(file).size();
    int expect[R];
    for (int r = 0; r < R; r++) expect[r] = 8;
    for (int a = 3; a < argc; a++) {
      int r, s;
      if (sscanf(argv[a], "%d:%d", &r, &s) == 2 && r >= 0 && r < R) expect[r] = s;
    }
    for (int r = 0; r < R; r++)
      th.emplace_back([&, r] {
        Rank k;
        gate();
        k.init(r, R, 720, 720, false, 0xC1);
        CHECK(k.ctx.trials_from_disk);
        CHECK(!trial_pending(&k.ctx));  // a record from the disk is complete: no settling launch, no trial
        k.launches = 0;
        k.trace_samples(77);
        CHECK(k.launches == 1 && k.trials_seen == 0);  // the first request is ONE launch of the recorded kernel
        shape[0][r]          = k.ctx.last_shape;
        timed_launches[0][r] = k.bench(0, 20, 77, &shape[0][r], &one_kernel[0][r]);
      });
    for (auto& t : th) t.join();
    for (int r = 0; r < R; r++) {
      CHECK(timed_launches[0][r] == 20 && one_kernel[0][r]);
      if (shape[0][r] != expect[r]) printf("rank %d runs shape %d, expected %d\n", r, shape[0][r], expect[r]);
      CHECK(shape[0][r] == expect[r]);
    }
    CHECK(read_lines(file).size() == before);  // a warm run writes nothing
    printf("warm: %zu lines unchanged, shapes", before);
    for (int r = 0; r < R; r++) printf(" %d", shape[0][r]);
    printf("\n");
  }
  printf(fails ? "trial ranks: %d checks FAILED\n" : "trial ranks: all checks passed\n", fails.load());
  return fails ? 1 : 0;
}
