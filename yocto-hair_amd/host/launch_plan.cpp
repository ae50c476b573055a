// launch_plan.cpp — which kernel a launch runs (timing trials, their record per process and on disk) and the hand-out order
// of its work items.
#include "context_internal.h"

#include "build_id.h"  // YH_BUILD_ID: a hash of the device and host sources, written by the Makefile

// Dense or sparse? When every pixel is expensive the quad kernel is latency-bound and more waves per SIMD pay
// (k_trace 256 x 5: C2 +19 %, C4 +10 % over 512 x 4); when a few expensive pixels bound the launch (C1: the hair covers
// 11 % of the frame and barely fills the resident waves) they cost 11 %. Measure: the number of
// max-cost work items the last launch was worth (sum of item costs over the largest) against the
// resident wave slots — measured 3 066 (C1), 4 777 (C4), 8 989 (C2), 24 970 (C3) against 4 096: with
// fewer expensive items than slots every wave that can run already does. This only picks the CANDIDATES; which
// kernel runs is measured (pick_launch_shape). YHAIR_SHAPE=0..3 overrides.
// Is the launch worth more max-cost work items than there are resident waves? (item costs of a k_trace launch)
static bool dense_by_costs(const yh_context* ctx, bool* known, bool* chain_bound = nullptr, bool* chain16 = nullptr) {
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
// One 32-sample trial is a noisy measurement (± 5 % launch to launch): when the runner-up is within YH_TRIAL_TIE of the
// best, both are tried a second time and the minimum of a kernel's trials counts, so that two ranks rendering halves of
// one image, or two renders of one image, do not settle on different kernels by chance.
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
static TrialKey trial_key(const yh_context* ctx) {
  return TrialKey{ctx->scene_key, ctx->state.width, ctx->state.height, ctx->rank, ctx->world, ctx->state.bounces};
}
// The same record ON DISK (round 4), so that the kernel an image runs does not depend on a handful of 32-sample launches
// re-decided by every process (every rank of every run). OPT-IN since round 5 — a library that writes files its host application did
// not ask for is a side effect: yh_set_trial_cache_dir(dir) (the CLIs and bench.py call it with yh_default_trial_cache_dir() =
// $XDG_CACHE_HOME/yhair or ~/.cache/yhair) or the environment's YHAIR_CACHE_DIR name the directory; without either nothing is read or
// written. YHAIR_NO_DISK_CACHE (or YHAIR_NO_TRIAL_CACHE, which also forgets the per-process record) switches it off whatever was set.
// File: <dir>/trials_v2.txt, one line per record, keyed by device (name + CU count), a fingerprint of the LOADED library file (so that
// builds with other flags — tools/build_variants.sh — never share records), YHAIR_DEVICE_SHARE and the TrialKey; appended with one
// O_APPEND write (atomic between the ranks of a run), the last line of a key counts. Only COMPLETE records are written (no candidate
// still wants a trial) and a loaded one is complete by construction, so a process that finds its image here runs no trial at all.
static std::mutex  g_cache_dir_mutex;
static std::string g_cache_dir;  // yh_set_trial_cache_dir
const char* yh_default_trial_cache_dir(void) {
  static std::string dir = [] {
    if (const char* x = getenv("XDG_CACHE_HOME")) return std::string(x) + "/yhair";
    if (const char* h = getenv("HOME")) return std::string(h) + "/.cache/yhair";
    return std::string();
  }();
  return dir.c_str();
}
int yh_set_trial_cache_dir(const char* dir) {
  std::lock_guard<std::mutex> lock(g_cache_dir_mutex);
  g_cache_dir = dir ? dir : "";
  return YH_OK;
}
static std::string disk_cache_path() {
  if (getenv("YHAIR_NO_DISK_CACHE") || getenv("YHAIR_NO_TRIAL_CACHE")) return "";
  std::string dir;
  if (const char* e = getenv("YHAIR_CACHE_DIR")) dir = e;
  else {
    std::lock_guard<std::mutex> lock(g_cache_dir_mutex);
    dir = g_cache_dir;
  }
  if (dir.empty()) return "";
  return dir + "/trials_v2.txt";
}
// FNV-1a over the shared library this code was loaded from (found through dladdr): device code, host code and every -D of the build
// in one fingerprint. Computed once per process (5 MB: a few milliseconds); the Makefile's source hash when the file cannot be read.
static const std::string& library_fingerprint() {
  static const std::string fp = [] {
    Dl_info info{};
    if (dladdr((const void*)&yh_set_trial_cache_dir, &info) && info.dli_fname) {
      if (FILE* f = fopen(info.dli_fname, "rb")) {
        uint64_t h = 1469598103934665603ull;
        unsigned char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0)
          for (size_t i = 0; i < n; i++) h = (h ^ buf[i]) * 1099511628211ull;
        fclose(f);
        char out[32];
        snprintf(out, sizeof(out), "%016llx", (unsigned long long)h);
        return std::string(out);
      }
    }
    return std::string(YH_BUILD_ID);
  }();
  return fp;
}
static std::string disk_key(const yh_context* ctx) {
  const char* share = getenv("YHAIR_DEVICE_SHARE");
  char buf[320];
  snprintf(buf, sizeof(buf), "%s|%s|%s|%016llx|%d|%d|%d|%d|%d", ctx->device_name.c_str(), library_fingerprint().c_str(), share ? share : "1",
      (unsigned long long)ctx->scene_key, ctx->state.width, ctx->state.height, ctx->rank, ctx->world, ctx->state.bounces);
  return buf;
}
static void mkdirs(const std::string& file) {
  for (size_t i = 1; i < file.size(); i++)
    if (file[i] == '/') (void)mkdir(file.substr(0, i).c_str(), 0755);
}
static void disk_store(const yh_context* ctx, const TrialRecord& r) {
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
static bool disk_load(const yh_context* ctx, TrialRecord& r) {
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
      if (ok && t.ms[k] == -1.0) t.ms[k] = std::numeric_limits<double>::infinity();  // a candidate that cannot run on this device
      else if (ok && !(std::isfinite(t.ms[k]) && t.ms[k] >= 0 && t.trials[k] >= 0 && (t.ms[k] > 0) == (t.trials[k] > 0))) ok = false;  // a damaged line: not a record
    }
    if (ok && sscanf(p, " ; %d %d %d", &t.dense, &t.chain, &t.chain16) == 3) r = t, found = true;  // (the last line of a key counts)
  }
  fclose(f);
  return found;
}
static void trials_store(const yh_context* ctx) {
  TrialRecord r;
  for (int k = 0; k < YH_SHAPES; k++) r.ms[k] = ctx->shape_ms[k], r.trials[k] = ctx->shape_trials[k];
  r.dense = ctx->dense, r.chain = ctx->chain, r.chain16 = ctx->chain16;
  {
    std::lock_guard<std::mutex> lock(g_trials_mutex);
    g_trials[trial_key(ctx)] = r;
  }
  // on disk only what a later process may rely on: the choice was made by the trials (no forced shape, no heuristic-only
  // mode), dense / sparse is known, and nothing is left to try on this image
  // — and once per image and context: later trial-length launches refine the times in memory, they do not grow the file
  if (!ctx->trials_on_disk && !trials_off() && ctx->dense >= 0 && ctx->costs_settled && !trial_pending(ctx)) {
    disk_store(ctx, r);
    const_cast<yh_context*>(ctx)->trials_on_disk = true;
  }
}
void trials_load(yh_context* ctx) {
  ctx->trials_from_disk = false, ctx->trials_on_disk = false;
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
    ctx->trials_from_disk = true, ctx->trials_on_disk = true;
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
static int candidates(const yh_context* ctx, int cand[6]) {
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
static bool wants_trial(const yh_context* ctx, const int* cand, int n, int c) {
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
// The octet kernel (launch shape 4: eight lanes per path) takes HALF a quadrant per wave: entry = item << 1 | half, the two
// halves of an item next to each other in the cost-sorted order.
void split_items_for_octets(std::vector<int>& items) {
  std::vector<int> out;
  out.reserve(items.size() * 2);
  for (int it : items) out.push_back(it << 1), out.push_back((it << 1) | 1);
  items.swap(out);
}
// ... and the sixteen-lane form (shape 6) a QUARTER: entry = item << 2 | row of the 4x4 block.
void split_items_for_hex(std::vector<int>& items) {
  std::vector<int> out;
  out.reserve(items.size() * 4);
  for (int it : items)
    for (int k = 0; k < 4; k++) out.push_back((it << 2) | k);
  items.swap(out);
}
int upload_work_items(yh_context* ctx) {
  std::vector<int> tiles;
  build_work_items(ctx, tiles);
  ctx->state.num_groups = 1, ctx->state.group_begin[0] = 0, ctx->state.group_begin[1] = (int)tiles.size();
  if (ctx->state.shader == YH_SHADER_PATH && ctx->state.launch_shape == 3) deal_items_for_stream(ctx, tiles);
  if (ctx->state.shader == YH_SHADER_PATH && (ctx->state.launch_shape == 4 || ctx->state.launch_shape == 7)) split_items_for_octets(tiles);
  if (ctx->state.shader == YH_SHADER_PATH && ctx->state.launch_shape == 5) split_items_side_by_side(ctx, tiles);
  if (ctx->state.shader == YH_SHADER_PATH && (ctx->state.launch_shape == 6 || ctx->state.launch_shape == 8)) split_items_for_hex(tiles);
  lay_out_first_round(ctx, tiles, ctx->state.shader == YH_SHADER_PATH ? ctx->state.launch_shape : 0);
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
int expensive_items(const yh_context* ctx, const std::vector<int>& items) {
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
bool side_by_side_grids(const yh_context* ctx, int* oct_blocks, int* quad_blocks) {
  const int lds = yhk_trace_sbs_lds_bytes(&ctx->scene), occ = yhk_trace_sbs_occupancy(lds, ctx->scene.general_materials);
  if (occ < 1) return false;
  const int resident = ctx->num_cus * occ;
  *oct_blocks  = std::min((ctx->hy_oct_entries + 7) / 8, resident / 2);  // (the quad workgroups keep at least half of the device)
  *quad_blocks = ctx->hy_quad_items > 0 ? std::max(1, std::min((ctx->hy_quad_items + 7) / 8, resident - *oct_blocks)) : 0;
  return true;
}
void split_items_side_by_side(yh_context* ctx, std::vector<int>& items) {
  // How many: the launch of a sparse image ends with a handful of quadrants that are the most expensive ones in EVERY launch
  // (C1: sixteen widened items take 10 % off the launch, 128 no more, 512 lose it again to the extra waves —
  // profiles/r03/side_by_side_fused_ab.txt): a sixty-fourth of the expensive items, sixteen at least.
  const int H = expensive_items(ctx, items);
  int n_oct = std::min(H / 2, std::max(16, std::min(256, H / 64)));
  std::vector<int> out;
  out.reserve(items.size() + n_oct);
  for (size_t i = (size_t)n_oct; i < items.size(); i++) out.push_back(items[i]);                   // quads: the rest, most expensive first
  for (int i = 0; i < n_oct; i++) out.push_back(items[i] << 1), out.push_back((items[i] << 1) | 1);  // octets: two half-quadrant entries each
  ctx->hy_quad_items = (int)items.size() - n_oct, ctx->hy_oct_entries = 2 * n_oct;
  ctx->hy_oct_items.assign(items.begin(), items.begin() + n_oct);
  items.swap(out);
  {  // both lists by wave slot: the octet workgroups are the first of the launch, the quad ones follow (side_by_side_impl)
    int G_o = 0, G_q = 0;
    side_by_side_grids(ctx, &G_o, &G_q);
    if (G_o > 0) lay_out_range(ctx, items.data() + ctx->hy_quad_items, (size_t)ctx->hy_oct_entries, 8, G_o, 0);
    if (G_q > 0) lay_out_range(ctx, items.data(), (size_t)ctx->hy_quad_items, 8, G_q, G_o);
  }
}
// THE HEAD OF THE LIST BY POSITION. A wave of k_trace takes its first item from the list entry at its own position
// (workgroup x waves per workgroup + wave; csrc/dev_items.h) and later ones from the cursor behind those positions. The four
// wave slots of a SIMD do not run at the same speed: on C1 the same kind of item takes 10.4 ms in hardware slot 0, 11.0 in
// slot 1, 11.8 in slot 2 and 13.3 in slot 3 (profiles/r03/where_items_ran.txt: the issue arbiter favours the older wave), and
// the launch ends with its slowest item. A wave's slot follows from the dispatch order: the workgroups come round by round,
// one per CU and round, and waves w and w + 4 of a 512-thread workgroup share a SIMD — so slot = round x (waves per workgroup
// / 4) + wave / 4. The most expensive items go to the slot-0 waves, the next to slot 1, and so on: on a sparse image the
// slowest slot holds none of the expensive items. Purely a matter of time: whatever the layout, every entry is taken once.
void lay_out_range(const yh_context* ctx, int* items, size_t n, int wpb, int G, int block_offset) {  // entries [0, n) of one list, its G workgroups; block_offset: workgroups of the same launch dispatched before them
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
void lay_out_first_round(const yh_context* ctx, std::vector<int>& items, int shape) {
  if (shape == 3 || shape == 5 || items.empty()) return;  // (k_stream deals its items itself; side by side lays its two lists out when it splits them)
  const int wpb = yhk_block_threads(shape) / 64;
  const int occ = yhk_trace_occupancy(yhk_trace_lds_bytes(&ctx->scene, shape), ctx->scene.general_materials, shape);
  if (occ < 1 || wpb < 1) return;
  const int G = std::max(1, std::min(((int)items.size() + wpb - 1) / wpb, ctx->num_cus * occ));  // the grid trace_impl launches
  lay_out_range(ctx, items.data(), items.size(), wpb, G);
}

// Bookkeeping after a synchronous launch: its time (kernel selection) and, after launches 1, 2, 4, 8, ... of a state,
// the longest-processing-time-first order for the next ones (the pixel results do not depend on either).
int replan_after_launch(yh_context* ctx, int nsamples) {
  // A pixel's samples are sequential, so the items that start last bound the launch; hair quadrants
  // cost 10-100x background ones. Re-planned after launches 1, 2, 4, 8, ... of a state: the relative
  // costs of the items settle after the first launches (they are a property of the IMAGE — the count goes on over
  // a yh_init_state of the same image, round 5: the bench's timed region re-planned five times in twenty steps otherwise),
  // and the read-back, sort and upload are half a millisecond of a 16 ms launch.
  const unsigned li      = ++ctx->launches_of_image;
  // (... and after the first launch long enough to settle the costs, whenever it comes: the kernel trials wait for it)
  // (... and after EVERY launch long enough for the half millisecond to be under one per cent of it: k_stream's per-wave shares are corrected by the re-plan,
  // and a dense image's launches take a hundred times what it costs)
  const bool     refresh = (li & (li - 1)) == 0 || ctx->last_ms >= 50.0f || (!ctx->costs_settled && nsamples >= YH_TRIAL_SPP && ctx->state.shader == YH_SHADER_PATH);
  if (refresh) HIPCHK(ctx, hipMemcpy(ctx->item_cost.data(), ctx->d_tile_cost.p, ctx->item_cost.size() * 4, hipMemcpyDeviceToHost));
  if (refresh) ctx->last_nsamples = nsamples;
  if (refresh && ctx->last_shape == 5)  // an item that ran as octets reports the time of its two halves, 2 x 0.74 of what it costs as a quad
    for (int it : ctx->hy_oct_items) ctx->item_cost[(size_t)it] = (unsigned int)((double)ctx->item_cost[(size_t)it] * (1.0 / 1.48));
  if (ctx->state.shader == YH_SHADER_PATH && (refresh || ctx->have_costs)) record_launch(ctx, nsamples, refresh);
  if (!refresh) return YH_OK;
  return upload_work_items(ctx);
}

// The one-lane kernels (k_stream, k_intersect_lanes) read the lane blob through buffer loads with 32-bit byte offsets (csrc/dev_lane.h): the
// part of it they read — test records and 4-wide nodes — has to end below 4 GB (about fifty million hair segments; the BASELINE configs hold 0.4-3.2 M).
bool lane_kernels_can_address(const yh_context* ctx) { return (long long)ctx->lane_units * 32ll <= 0xFFFFFE00ll; }

// Launch geometry of the streaming integrator: path slots per wave and workgroups. The pixels of the launch are
// spread over as many waves as the CUs hold, each wave with a few paths per lane so that its lanes stay full
// between stages: 128 .. 192 slots (more waves beat fuller batches: measured on C2 / C3, profiles/r02;
// YHAIR_ST_SLOTS / YHAIR_ST_WAVES: developer switches). Returns 0 when the kernel cannot run.
int stream_geometry(const yh_context* ctx, int num_items, int* slots_per_wave, int* grid_blocks, int* lds_out, bool* single_generation) {
  if (!lane_kernels_can_address(ctx)) return 0;  // (the caller drops the candidate: the quad kernels render the same bits)
  const int     wpb    = yhk_stream_block_threads() / 64;
  const int64_t pixels = (int64_t)num_items * 16;  // work items are 4x4 pixel quadrants
  int           P      = (int)std::max<int64_t>(128, std::min<int64_t>(192, (pixels / ((int64_t)ctx->num_cus * 16) + 63) / 64 * 64));
  const bool    forced = getenv("YHAIR_ST_SLOTS") != nullptr;
  if (forced) P = std::max(64, std::min(4096, atoi(getenv("YHAIR_ST_SLOTS")) / 64 * 64));
  int lds_bytes = yhk_stream_lds_bytes(YHD_LDS_TABLES_F4(&ctx->scene), P);
  int occupancy = yhk_stream_occupancy(lds_bytes, ctx->scene.general_materials);
  if (occupancy < 1) return 0;
  if (const char* env = getenv("YHAIR_ST_WAVES")) occupancy = std::max(1, std::min(occupancy, (atoi(env) + wpb - 1) / wpb));  // waves per CU
  int64_t want = (pixels + (int64_t)P * wpb - 1) / ((int64_t)P * wpb);
  int     grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * occupancy));
  // ONE GENERATION (round 5): when the resident waves hold the whole image at once (C2 at 720^2: 127 pixels per wave) no wave ever takes a
  // second helping, so nothing evens out that the four hardware wave slots of a SIMD run at different speeds — with equal shares the
  // slot-0 waves are done at 0.69 of the launch and the slot-3 waves at 0.86-1.0 (profiles/r05/k_stream_wave_shares.txt). Then every
  // resident wave is used, a wave's share of the pixels follows its slot's speed (deal_items_for_stream), and the pool has room for the
  // largest share: 1.25 x the average, in 64-slot steps.
  bool one = false;
  if (!forced) {
    const int64_t waves_max = (int64_t)ctx->num_cus * occupancy * wpb, per_wave = (pixels + waves_max - 1) / std::max<int64_t>(1, waves_max);
    if (pixels >= 64 * waves_max && per_wave <= 204) {
      const int P1   = (int)std::min<int64_t>(256, (per_wave * 5 / 4 + 63) / 64 * 64);
      const int lds1 = yhk_stream_lds_bytes(YHD_LDS_TABLES_F4(&ctx->scene), P1);
      if (P1 >= P && yhk_stream_occupancy(lds1, ctx->scene.general_materials) >= occupancy) P = P1, lds_bytes = lds1, grid = (int)(waves_max / wpb), one = true;
    }
  }
  *slots_per_wave = P;
  *grid_blocks    = grid;
  if (lds_out) *lds_out = lds_bytes;
  if (single_generation) *single_generation = one;
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
// SHARES BY WAVE-SLOT SPEED (round 5; stream_geometry's "one generation" case). Wave w of the launch starts with the entries
// [wave_begin[w], wave_begin[w + 1]) of the list (csrc/stream.hip); the items are dealt out longest-processing-time first — most expensive
// item to the wave with the largest remaining budget — where a wave's budget is the image's total cost times the relative speed of its
// hardware wave slot (= the dispatch round of its workgroup: one workgroup per CU and round; yh_context::stream_speed, measured by every
// k_stream launch from the waves' own begin / end stamps: note_stream_wave_log). Pixel results do not depend on who renders them.
static bool deal_shares_by_speed(yh_context* ctx, std::vector<int>& items, int P, int grid) {
  const int    wpb = yhk_stream_block_threads() / 64, cap = P / 16;
  const size_t waves = (size_t)grid * wpb, n = items.size();
  if (n > waves * (size_t)cap) return false;
  auto round_of = [&](size_t w) { return (w / wpb) / (size_t)std::max(1, ctx->num_cus); };
  if (ctx->item_scale.size() != ctx->item_cost.size()) ctx->item_scale.assign(ctx->item_cost.size(), 1.0f);
  // FEEDBACK from the last launch that ran on shares (its waves' begin / end stamps, note_stream_wave_log): what a dispatch round gets
  // through per tick — planned cost over time, blended into stream_speed — and, per wave, how much longer or shorter it took than
  // its round's rate says: its items' costs are corrected by that ratio (the BVH steps an item reports are not all of its time: the same
  // waves end late launch after launch while the plan stands, correlation 0.96, and others do after a re-deal).
  if (ctx->st_log_fresh && ctx->st_share_begin.size() == waves + 1 && ctx->st_last_log.size() == 2 * waves) {
    const size_t rounds = round_of(waves - 1) + 1;
    std::vector<double> planned(waves, 0.0), took(waves, 0.0), pr(rounds, 0.0), tr(rounds, 0.0);
    for (size_t w = 0; w < waves; w++) {
      for (int k = ctx->st_share_begin[w]; k < ctx->st_share_begin[w + 1]; k++) planned[w] += ctx->st_share_cost[(size_t)k];
      const unsigned long long* e = &ctx->st_last_log[2 * w];
      took[w] = e[1] > e[0] ? (double)(e[1] - e[0]) : 0.0;
      if (took[w] > 0 && planned[w] > 0) pr[round_of(w)] += planned[w], tr[round_of(w)] += took[w];
    }
    bool all = true;
    double mean = 0;
    for (size_t r = 0; r < rounds; r++) all = all && tr[r] > 0, mean += all ? pr[r] / tr[r] : 0.0;
    if (all) {
      mean /= (double)rounds;
      if (ctx->stream_speed.size() != rounds) ctx->stream_speed.assign(rounds, 1.0);
      for (size_t r = 0; r < rounds; r++) ctx->stream_speed[r] = 0.5 * ctx->stream_speed[r] + 0.5 * (pr[r] / tr[r]) / mean;
      if (ctx->st_wave_speed.size() != waves) ctx->st_wave_speed.assign(waves, 1.0f);
      for (size_t w = 0; w < waves; w++) {
        if (!(took[w] > 0 && planned[w] > 0)) continue;
        // what is left after the round's rate and the wave's own factor so far: half of it goes to the wave's items, a third to the wave's
        // position (a CU or an XCC that runs slower shows up launch after launch whatever it is given; an item's deficit moves with the item)
        const double expected = planned[w] / ((pr[round_of(w)] / tr[round_of(w)]) * (double)ctx->st_wave_speed[w]);
        const double ratio    = std::min(1.5, std::max(0.67, took[w] / expected));
        const float  corr     = (float)std::pow(ratio, 0.5);
        for (int k = ctx->st_share_begin[w]; k < ctx->st_share_begin[w + 1]; k++) {
          if (ctx->st_share_items[(size_t)k] < 0) continue;  // padding
          float& sc = ctx->item_scale[(size_t)ctx->st_share_items[(size_t)k]];
          sc        = std::min(4.0f, std::max(0.25f, sc * corr));
        }
        ctx->st_wave_speed[w] = std::min(1.5f, std::max(0.67f, ctx->st_wave_speed[w] * (float)std::pow(ratio, -0.33)));
      }
      if (getenv("YHAIR_TIMING")) {
        fprintf(stderr, "[yhair] k_stream shares: wave-slot speeds (dispatch rounds)");
        for (double v : ctx->stream_speed) fprintf(stderr, " %.3f", v);
        fprintf(stderr, "\n");
      }
    }
  }
  ctx->st_log_fresh = false;
  std::vector<std::pair<double, int>> by_cost(n);  // (cost, item), most expensive first
  double total = 0;
  for (size_t i = 0; i < n; i++) {
    by_cost[i] = {((double)ctx->item_cost[(size_t)items[i]] + 1.0) * (double)ctx->item_scale[(size_t)items[i]], items[i]};
    total += by_cost[i].first;
  }
  std::sort(by_cost.begin(), by_cost.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
  double ssum = 0;
  auto speed_of = [&](size_t w) {
    return (ctx->stream_speed.empty() ? 1.0 : ctx->stream_speed[std::min(round_of(w), ctx->stream_speed.size() - 1)]) * (ctx->st_wave_speed.size() == waves ? (double)ctx->st_wave_speed[w] : 1.0);
  };
  for (size_t w = 0; w < waves; w++) ssum += speed_of(w);
  // (flat arrays, `cap` places per wave: this runs after every launch of a dense image, and four thousand pairs of small vectors were most of its five milliseconds)
  std::vector<int>      share_items(waves * (size_t)cap);
  std::vector<double>   share_cost(waves * (size_t)cap);
  std::vector<int>      share_n(waves, 0);
  std::vector<std::pair<double, uint32_t>> heap(waves);  // (remaining budget, wave): max-heap
  for (size_t w = 0; w < waves; w++) heap[w] = {total * speed_of(w) / ssum, (uint32_t)w};
  std::make_heap(heap.begin(), heap.end());
  for (size_t i = 0; i < n; i++) {
    if (heap.empty()) return false;  // (cannot happen: n <= waves x cap)
    std::pop_heap(heap.begin(), heap.end());
    auto top = heap.back();
    heap.pop_back();
    const size_t at = (size_t)top.second * (size_t)cap + (size_t)share_n[top.second]++;
    share_items[at] = by_cost[i].second, share_cost[at] = by_cost[i].first;
    top.first -= by_cost[i].first;
    if (share_n[top.second] < cap) heap.push_back(top), std::push_heap(heap.begin(), heap.end());
  }
  std::vector<int> begin(waves + 1, 0), out;
  std::vector<double> out_cost;
  out.reserve(n + 3 * waves), out_cost.reserve(n + 3 * waves);
  for (size_t w = 0; w < waves; w++) {
    begin[w] = (int)out.size();
    out.insert(out.end(), share_items.begin() + w * cap, share_items.begin() + w * cap + share_n[w]);
    out_cost.insert(out_cost.end(), share_cost.begin() + w * cap, share_cost.begin() + w * cap + share_n[w]);
    while ((out.size() - (size_t)begin[w]) % 4 != 0) out.push_back(-1), out_cost.push_back(0.0);  // a take is four entries (csrc/stream.hip): -1 = no item
  }
  begin[waves] = (int)out.size();
  if (upload_keep(ctx, ctx->d_st_wave_begin, begin.data(), begin.size() * 4) != YH_OK) return false;
  ctx->stream_pool.wave_begin = (const int*)ctx->d_st_wave_begin.p;
  ctx->st_share_waves         = waves, ctx->st_share_slots = P;
  ctx->st_share_begin = begin, ctx->st_share_items = out, ctx->st_share_cost = out_cost;  // what the next feedback reads the launch against
  items.swap(out);
  ctx->state.num_groups = 1, ctx->state.group_begin[0] = (int)items.size(), ctx->state.group_begin[1] = (int)items.size();  // nothing is left for the cursor
  return true;
}
// After a synchronous k_stream launch: keep the waves' stamps for the next hand-out (deal_shares_by_speed reads them once).
void note_stream_wave_log(yh_context* ctx, const unsigned long long* log, size_t waves) {
  ctx->st_last_log.assign(log, log + 2 * waves);
  ctx->st_log_fresh = ctx->stream_pool.wave_begin != nullptr && ctx->st_share_waves == waves;  // (only a launch that ran on shares says something about them)
}

void deal_items_for_stream(yh_context* ctx, std::vector<int>& items) {
  int  P = 0, grid = 0;
  bool one = false;
  ctx->state.num_groups = 1, ctx->state.group_begin[0] = 0, ctx->state.group_begin[1] = (int)items.size();
  ctx->stream_pool.wave_begin = nullptr, ctx->st_share_waves = 0, ctx->st_share_slots = 0;
  ctx->st_items = (int)items.size();  // the geometry of k_stream's launches follows the ITEMS of the list, not its entries (a shared-out list is padded)
  if (items.empty() || !stream_geometry(ctx, (int)items.size(), &P, &grid, nullptr, &one)) return;
  if (one && deal_shares_by_speed(ctx, items, P, grid)) return;
  const size_t R = (size_t)grid * (yhk_stream_block_threads() / 64) * (size_t)(P / 64);  // takes resident together
  // (Groups — one compact image region per XCD, the items in Morton order cut into runs of equal cost — were measured without gain in
  // round 2, profiles/r02/k_stream_xcd_groups.txt: C3 296 -> 300 Msamples/s, C2 237 -> 222; after the first bounce the rays of a region
  // wander through the hair, and 4 MB of L2 hold little of a region's 40 MB anyway. The device side still takes groups; the host deals one.)
  std::vector<int> out;
  out.reserve(items.size());
  deal_block(out, items.data(), items.size(), R);
  items.swap(out);
}
