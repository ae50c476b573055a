// bvh_gpu.hip — the reference's BVH (build_bvh / split_middle, pt.cpp:557-650) built on the GPU,
// node for node and primitive for primitive.
//
// The tree cannot be an LBVH: leaf order decides which primitive wins an exact-distance tie
// (math.h:3450), so the device build reproduces the reference's construction level by level:
//   * a node's split (largest axis of the centroid bounds, middle of that axis) depends only on the
//     set of primitives in its range: centroid bounds by (wave-reduced) atomic min / max on
//     order-preserving integer keys — exact;
//   * std::partition (libstdc++, bidirectional form) swaps the k-th element that fails the
//     predicate, counted from the left, with the k-th element that passes it, counted from the
//     right, for as long as the first lies before the second. With one exclusive scan of the
//     predicate flags every element knows its k and where its partner is, so the whole level is
//     permuted in one pass into exactly the order the sequential algorithm leaves;
//   * nodes are numbered as the reference's breadth-first queue numbers them: a level's internal
//     nodes, left to right, receive consecutive pairs of children.
// Node boxes are formed bottom-up afterwards (leaf: union of its primitives' boxes in leaf order;
// internal node: union of its two children), which gives the reference's boxes bit for bit.
// tests/test_gpu_parity.py compares the result with the host builder (itself checked against the
// oracle's tree) on every shape of the golden scenes and at the full 1.6 M-segment size.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <string.h>

namespace {

__device__ __forceinline__ unsigned int fkey(float f) {  // order-preserving float -> uint
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned int k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
__device__ __forceinline__ float fmin_(float a, float b) { return (a < b) ? a : b; }  // math.h:1779
__device__ __forceinline__ float fmax_(float a, float b) { return (a > b) ? a : b; }

// An open segment of the current level: primitives [start, end) belong to node `node`.
struct Seg {
  int start, end, node, pad;
};
// Per-segment scratch of one level
struct SegWork {
  unsigned int kmin[3], kmax[3];  // centroid bounds as keys
  int          split;             // 1: internal node
  int          nopart;            // 1: degenerate bounds, no partition (pt.cpp:577)
  int          axis;
  float        middle;
  int          n_true, mid, rank;
};

__global__ void k_centers(int n, const float* boxes, float* cx, float* cy, float* cz, int* pid, int* seg) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = boxes + 6 * (size_t)i;
  cx[i] = (b[0] + b[3]) / 2, cy[i] = (b[1] + b[4]) / 2, cz[i] = (b[2] + b[5]) / 2;  // center(bbox), math.h:3008
  pid[i] = i;
  seg[i] = 0;
}
__global__ void k_seg_init(int m, const Seg* segs, SegWork* work) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= m) return;
  SegWork w;
  for (int k = 0; k < 3; k++) w.kmin[k] = 0xFFFFFFFFu, w.kmax[k] = 0u;
  w.split = (segs[s].end - segs[s].start) > 4;  // bvh_max_prims (pt.cpp:598)
  w.nopart = 0, w.axis = 0, w.middle = 0, w.n_true = 0, w.mid = 0, w.rank = 0;
  work[s] = w;
}
// centroid bounds of the splitting segments: one atomic per wave when the wave lies in one segment
__global__ void k_bounds(int n, const int* seg, const float* cx, const float* cy, const float* cz, SegWork* work) {
  int  i = blockIdx.x * blockDim.x + threadIdx.x;
  int  s = i < n ? seg[i] : -1;
  bool on = s >= 0 && work[s].split;
  unsigned int k[6];
  if (on) {
    k[0] = fkey(cx[i]), k[1] = fkey(cy[i]), k[2] = fkey(cz[i]);
    k[3] = k[0], k[4] = k[1], k[5] = k[2];
  }
  int first = __shfl(s, 0, 64);
  if (__all(s == first)) {  // whole wave in one segment (always true near the top of the tree)
    if (!on) return;
    for (int off = 32; off > 0; off >>= 1)
      for (int c = 0; c < 3; c++) {
        k[c]     = min(k[c], (unsigned int)__shfl_xor((int)k[c], off, 64));
        k[3 + c] = max(k[3 + c], (unsigned int)__shfl_xor((int)k[3 + c], off, 64));
      }
    if ((threadIdx.x & 63) == 0)
      for (int c = 0; c < 3; c++) atomicMin(&work[s].kmin[c], k[c]), atomicMax(&work[s].kmax[c], k[3 + c]);
  } else if (on) {
    for (int c = 0; c < 3; c++) atomicMin(&work[s].kmin[c], k[c]), atomicMax(&work[s].kmax[c], k[3 + c]);
  }
}
// split_middle's choice of axis and position (pt.cpp:564-585)
__global__ void k_decide(int m, SegWork* work) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= m || !work[s].split) return;
  SegWork& w = work[s];
  float cmin[3], cmax[3], cs[3];
  for (int k = 0; k < 3; k++) cmin[k] = fkey_inv(w.kmin[k]), cmax[k] = fkey_inv(w.kmax[k]), cs[k] = cmax[k] - cmin[k];
  int axis = 0;
  if (cs[0] == 0 && cs[1] == 0 && cs[2] == 0) {
    w.nopart = 1, w.axis = 0;
    return;
  }
  if (cs[0] >= cs[1] && cs[0] >= cs[2]) axis = 0;
  if (cs[1] >= cs[0] && cs[1] >= cs[2]) axis = 1;
  if (cs[2] >= cs[0] && cs[2] >= cs[1]) axis = 2;
  w.axis = axis, w.middle = (cmin[axis] + cmax[axis]) / 2;
}
__global__ void k_flags(int n, const int* seg, const SegWork* work, const float* cx, const float* cy, const float* cz,
    unsigned int* flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  unsigned int f = 0;
  if (i < n) {
    int s = seg[i];
    if (s >= 0 && work[s].split && !work[s].nopart) {
      int   a = work[s].axis;
      float c = a == 0 ? cx[i] : (a == 1 ? cy[i] : cz[i]);
      f       = c < work[s].middle ? 1u : 0u;
    }
  }
  flag[i] = f;  // flag[n] = 0: the scan then has n + 1 entries
}
__global__ void k_seg_counts(int m, const Seg* segs, SegWork* work, const unsigned int* scan) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= m || !work[s].split) return;
  SegWork& w   = work[s];
  int      st  = segs[s].start, en = segs[s].end;
  int      nt  = w.nopart ? 0 : (int)(scan[en] - scan[st]);
  int      mid = st + nt;  // what std::partition returns
  w.n_true     = nt;
  if (w.nopart || mid == st || mid == en) mid = (st + en) / 2;  // pt.cpp:577,591
  w.mid = mid;
}
// where the k-th failing element from the left and the k-th passing element from the right are
__global__ void k_positions(int n, const int* seg, const Seg* segs, const SegWork* work, const unsigned int* flag,
    const unsigned int* scan, int* fpos, int* tpos) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = seg[i];
  if (s < 0 || !work[s].split || work[s].nopart) return;
  int st = segs[s].start, en = segs[s].end;
  if (flag[i]) {
    int r = (int)(scan[en] - scan[i + 1]);  // passing elements to the right of i
    tpos[st + r] = i;
  } else {
    int k = (i - st) - (int)(scan[i] - scan[st]);  // failing elements to the left of i
    fpos[st + k] = i;
  }
}
__global__ void k_permute(int n, const int* seg, const Seg* segs, const SegWork* work, const unsigned int* flag,
    const unsigned int* scan, const int* fpos, const int* tpos, const float* cx, const float* cy, const float* cz,
    const int* pid, float* ox, float* oy, float* oz, int* opid) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int dest = i;
  int s    = seg[i];
  if (s >= 0 && work[s].split && !work[s].nopart) {
    int st = segs[s].start, en = segs[s].end;
    int nt = work[s].n_true, nf = (en - st) - nt;
    if (flag[i]) {
      int r = (int)(scan[en] - scan[i + 1]);
      if (r < nf && fpos[st + r] < i) dest = fpos[st + r];
    } else {
      int k = (i - st) - (int)(scan[i] - scan[st]);
      if (k < nt && i < tpos[st + k]) dest = tpos[st + k];
    }
  }
  ox[dest] = cx[i], oy[dest] = cy[i], oz[dest] = cz[i], opid[dest] = pid[i];
}
__global__ void k_split_flags(int m, const SegWork* work, unsigned int* f) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s <= m) f[s] = (s < m && work[s].split) ? 1u : 0u;
}
// node records of this level + the next level's segments (children allocated breadth-first)
__global__ void k_children(int m, const Seg* segs, const SegWork* work, const unsigned int* rank, int nodes_so_far,
    float* nodes8, Seg* next) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= m) return;
  Seg    sg = segs[s];
  float* o  = nodes8 + 8 * (size_t)sg.node;
  if (work[s].split) {
    int r     = (int)rank[s];
    int child = nodes_so_far + 2 * r;
    o[6]      = __int_as_float(child);
    o[7]      = __int_as_float(2 | (1 << 16) | (work[s].axis << 24));
    next[2 * r]     = Seg{sg.start, work[s].mid, child, 0};
    next[2 * r + 1] = Seg{work[s].mid, sg.end, child + 1, 0};
  } else {
    o[6] = __int_as_float(sg.start);
    o[7] = __int_as_float(sg.end - sg.start);
  }
}
__global__ void k_assign(int n, int* seg, const SegWork* work, const unsigned int* rank) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = seg[i];
  if (s < 0) return;
  seg[i] = work[s].split ? 2 * (int)rank[s] + (i >= work[s].mid ? 1 : 0) : -1;
}
// boxes: leaves from their primitives in leaf order, then one level at a time towards the root
__global__ void k_boxes(int first, int count, float* nodes8, const float* boxes, const int* pid) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  float* o    = nodes8 + 8 * (size_t)(first + t);
  int    meta = __float_as_int(o[7]), start = __float_as_int(o[6]);
  float  mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
  float  mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  if (meta & 0x10000) {
    for (int c = 0; c < 2; c++) {
      const float* b = nodes8 + 8 * (size_t)(start + c);
      for (int k = 0; k < 3; k++) mn[k] = fmin_(mn[k], b[k]), mx[k] = fmax_(mx[k], b[3 + k]);
    }
  } else {
    int num = meta & 0xFFFF;
    for (int i = start; i < start + num; i++) {
      const float* b = boxes + 6 * (size_t)pid[i];
      for (int k = 0; k < 3; k++) mn[k] = fmin_(mn[k], b[k]), mx[k] = fmax_(mx[k], b[3 + k]);
    }
  }
  for (int k = 0; k < 3; k++) o[k] = mn[k], o[3 + k] = mx[k];
}

struct Buf {
  void* p = nullptr;
  ~Buf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T>
  T* as() {
    return (T*)p;
  }
};

}  // namespace

#define BVH_CHECK(x)                \
  do {                              \
    hipError_t e_ = (x);            \
    if (e_ != hipSuccess) return (int)e_; \
  } while (0)

// The builder proper, device in, device out: d_boxes = n x (min[3], max[3]); d_nodes_out receives the nodes (8 floats each: box, start,
// num | internal << 16 | axis << 24 — the layout of yh_bvh_build and, byte for byte, of the host's yhh::Node), d_pid_out the leaf order
// (n ints). On the host: the node count, the number of levels and level_first[l] = the first node of level l (breadth-first numbering:
// a level is a contiguous range; level_first[levels] = the node count). d_nodes_out must hold 2 n + 1 records, level_first 130 ints.
static int build_resident(int n, const float* d_boxes_in, float* d_nodes_out, int* d_pid_out, int* num_nodes, int* levels_out, int* level_first,
    hipStream_t stream) {
  const int    T  = 256;
  const size_t N  = (size_t)n;
  Buf d_c[2][3], d_pid[2], d_seg, d_flag, d_scan, d_fpos, d_tpos, d_segs[2], d_work, d_sflag, d_rank, d_tmp;
  for (int b = 0; b < 2; b++) {
    for (int k = 0; k < 3; k++) BVH_CHECK(d_c[b][k].alloc(N * 4));
    BVH_CHECK(d_pid[b].alloc(N * 4));
    BVH_CHECK(d_segs[b].alloc((N + 2) * sizeof(Seg)));
  }
  BVH_CHECK(d_seg.alloc(N * 4));
  BVH_CHECK(d_flag.alloc((N + 1) * 4));
  BVH_CHECK(d_scan.alloc((N + 1) * 4));
  BVH_CHECK(d_fpos.alloc(N * 4));
  BVH_CHECK(d_tpos.alloc(N * 4));
  BVH_CHECK(d_work.alloc((N + 2) * sizeof(SegWork)));
  BVH_CHECK(d_sflag.alloc((N + 2) * 4));
  BVH_CHECK(d_rank.alloc((N + 2) * 4));
  size_t tmp_bytes = 0;
  BVH_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_flag.as<unsigned int>(), d_scan.as<unsigned int>(), (int)(N + 1), stream));
  BVH_CHECK(d_tmp.alloc(tmp_bytes));
  int cur = 0;
  hipLaunchKernelGGL(k_centers, dim3((n + T - 1) / T), dim3(T), 0, stream, n, d_boxes_in, d_c[0][0].as<float>(),
      d_c[0][1].as<float>(), d_c[0][2].as<float>(), d_pid[0].as<int>(), d_seg.as<int>());
  Seg root{0, n, 0, 0};
  BVH_CHECK(hipMemcpyAsync(d_segs[0].p, &root, sizeof(Seg), hipMemcpyHostToDevice, stream));
  int m = 1, nodes_so_far = 1, levels = 0, segbuf = 0;
  level_first[0] = 0;
  while (m > 0) {
    if (levels >= 126) return (int)hipErrorInvalidValue;
    levels++;
    level_first[levels] = nodes_so_far;
    Seg*     segs = d_segs[segbuf].as<Seg>();
    SegWork* work = d_work.as<SegWork>();
    float *cx = d_c[cur][0].as<float>(), *cy = d_c[cur][1].as<float>(), *cz = d_c[cur][2].as<float>();
    float *ox = d_c[cur ^ 1][0].as<float>(), *oy = d_c[cur ^ 1][1].as<float>(), *oz = d_c[cur ^ 1][2].as<float>();
    dim3 gm((m + T - 1) / T), gn((n + T - 1) / T), gn1((n + 1 + T - 1) / T), gm1((m + 1 + T - 1) / T);
    hipLaunchKernelGGL(k_seg_init, gm, dim3(T), 0, stream, m, segs, work);
    hipLaunchKernelGGL(k_bounds, gn, dim3(T), 0, stream, n, d_seg.as<int>(), cx, cy, cz, work);
    hipLaunchKernelGGL(k_decide, gm, dim3(T), 0, stream, m, work);
    hipLaunchKernelGGL(k_flags, gn1, dim3(T), 0, stream, n, d_seg.as<int>(), work, cx, cy, cz, d_flag.as<unsigned int>());
    BVH_CHECK(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_flag.as<unsigned int>(), d_scan.as<unsigned int>(), (int)(N + 1), stream));
    hipLaunchKernelGGL(k_seg_counts, gm, dim3(T), 0, stream, m, segs, work, d_scan.as<unsigned int>());
    hipLaunchKernelGGL(k_positions, gn, dim3(T), 0, stream, n, d_seg.as<int>(), segs, work, d_flag.as<unsigned int>(),
        d_scan.as<unsigned int>(), d_fpos.as<int>(), d_tpos.as<int>());
    hipLaunchKernelGGL(k_permute, gn, dim3(T), 0, stream, n, d_seg.as<int>(), segs, work, d_flag.as<unsigned int>(),
        d_scan.as<unsigned int>(), d_fpos.as<int>(), d_tpos.as<int>(), cx, cy, cz, d_pid[cur].as<int>(), ox, oy, oz,
        d_pid[cur ^ 1].as<int>());
    hipLaunchKernelGGL(k_split_flags, gm1, dim3(T), 0, stream, m, work, d_sflag.as<unsigned int>());
    size_t need = 0;
    BVH_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_sflag.as<unsigned int>(), d_rank.as<unsigned int>(), m + 1, stream));
    if (need > tmp_bytes) return (int)hipErrorInvalidValue;  // m + 1 <= n + 1: cannot happen
    BVH_CHECK(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_sflag.as<unsigned int>(), d_rank.as<unsigned int>(), m + 1, stream));
    hipLaunchKernelGGL(k_children, gm, dim3(T), 0, stream, m, segs, work, d_rank.as<unsigned int>(), nodes_so_far,
        d_nodes_out, d_segs[segbuf ^ 1].as<Seg>());
    hipLaunchKernelGGL(k_assign, gn, dim3(T), 0, stream, n, d_seg.as<int>(), work, d_rank.as<unsigned int>());
    unsigned int splits = 0;
    BVH_CHECK(hipMemcpyAsync(&splits, d_rank.as<unsigned int>() + m, 4, hipMemcpyDeviceToHost, stream));
    BVH_CHECK(hipStreamSynchronize(stream));
    nodes_so_far += 2 * (int)splits;
    m = 2 * (int)splits;
    cur ^= 1, segbuf ^= 1;
  }
  // boxes, deepest level first
  for (int l = levels - 1; l >= 0; l--) {
    int first = level_first[l], count = level_first[l + 1] - first;
    if (l == levels - 1) count = nodes_so_far - first;
    if (count <= 0) continue;
    hipLaunchKernelGGL(k_boxes, dim3((count + T - 1) / T), dim3(T), 0, stream, first, count, d_nodes_out, d_boxes_in, d_pid[cur].as<int>());
  }
  BVH_CHECK(hipMemcpyAsync(d_pid_out, d_pid[cur].p, N * 4, hipMemcpyDeviceToDevice, stream));
  BVH_CHECK(hipStreamSynchronize(stream));  // (the scratch buffers go out of scope)
  BVH_CHECK(hipGetLastError());
  level_first[levels] = nodes_so_far;
  *num_nodes = nodes_so_far, *levels_out = levels;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The rest of yh_upload_scene's per-shape work ON THE DEVICE (round 6): primitive bounds and leaf-ordered records from the shape's raw
// vertex arrays, and the 4- / 8- / 16-wide collapses of the tree written straight into the traversal kernels' array (yh_device.h:
// yhd_scene::lane_blob) — so that a hair model crosses PCIe once, as the 60 MB of its vertex arrays, and nothing of its tree comes back.
// Each kernel restates a host function of host/scene_upload.cpp / host/bvh_build.cpp bit for bit (same operands, same operations).
// ---------------------------------------------------------------------------------------------------------------------------------
// line_bounds / triangle_bounds (math.h:3037-3044), radius 0.001 when the shape has none (sceneio.cpp:390)
__global__ void k_prim_boxes(int lines, int n, const float* pos, const float* radius, const int* idx, float* boxes) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float* o = boxes + 6 * (size_t)e;
  if (lines) {
    int   a = idx[2 * (size_t)e], b = idx[2 * (size_t)e + 1];
    float r0 = radius ? radius[a] : 0.001f, r1 = radius ? radius[b] : 0.001f;
    for (int k = 0; k < 3; k++) {
      float p0 = pos[3 * (size_t)a + k], p1 = pos[3 * (size_t)b + k];
      o[k] = fmin_(p0 - r0, p1 - r1), o[3 + k] = fmax_(p0 + r0, p1 + r1);
    }
  } else {
    const float* p0 = pos + 3 * (size_t)idx[3 * (size_t)e];
    const float* p1 = pos + 3 * (size_t)idx[3 * (size_t)e + 1];
    const float* p2 = pos + 3 * (size_t)idx[3 * (size_t)e + 2];
    for (int k = 0; k < 3; k++) o[k] = fmin_(p0[k], fmin_(p1[k], p2[k])), o[3 + k] = fmax_(p0[k], fmax_(p1[k], p2[k]));
  }
}
// leaf-ordered records (yh_device.h): a segment = {p0, r0}{p1, r1}{t0, element}{t1, 0}, a triangle = {p0, element}{p1}{p2}{n0}{n1}{n2}
__global__ void k_leaf_records(int lines, int n, const int* pid, const float* pos, const float* nrm, const float* radius, const int* idx,
    float4* out) {
  int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const int   e  = pid[slot];
  const float ew = __int_as_float(e);
  auto P = [&](int v) { return make_float3(pos[3 * (size_t)v], pos[3 * (size_t)v + 1], pos[3 * (size_t)v + 2]); };
  auto N = [&](int v) { return nrm ? make_float3(nrm[3 * (size_t)v], nrm[3 * (size_t)v + 1], nrm[3 * (size_t)v + 2]) : make_float3(0, 0, 0); };
  if (lines) {
    int    a = idx[2 * (size_t)e], b = idx[2 * (size_t)e + 1];
    float3 p0 = P(a), p1 = P(b), t0 = N(a), t1 = N(b);
    float4* r = out + 4 * (size_t)slot;
    r[0] = make_float4(p0.x, p0.y, p0.z, radius ? radius[a] : 0.001f), r[1] = make_float4(p1.x, p1.y, p1.z, radius ? radius[b] : 0.001f);
    r[2] = make_float4(t0.x, t0.y, t0.z, ew), r[3] = make_float4(t1.x, t1.y, t1.z, 0.0f);
  } else {
    int    a = idx[3 * (size_t)e], b = idx[3 * (size_t)e + 1], c = idx[3 * (size_t)e + 2];
    float3 p0 = P(a), p1 = P(b), p2 = P(c), n0 = N(a), n1 = N(b), n2 = N(c);
    float4* r = out + 6 * (size_t)slot;
    r[0] = make_float4(p0.x, p0.y, p0.z, ew), r[1] = make_float4(p1.x, p1.y, p1.z, 0.0f), r[2] = make_float4(p2.x, p2.y, p2.z, 0.0f);
    r[3] = make_float4(n0.x, n0.y, n0.z, 0.0f), r[4] = make_float4(n1.x, n1.y, n1.z, 0.0f), r[5] = make_float4(n2.x, n2.y, n2.z, 0.0f);
  }
}
// THE WIDE COLLAPSES (host/bvh_build.cpp: collapse_wide / _wide8 / _wide16). A W-wide node (W = 2^L) stands for L levels of the binary tree below
// a binary node at a level that is a multiple of L; the host numbers the wide nodes in its breadth-first queue's order, children in slot order.
// The binary tree is itself numbered breadth-first (a node's children are consecutive, allocated in the order of their parents), so the nodes of
// one level are in the order of their paths from the root — and the wide nodes, which all sit at levels 0, L, 2 L, ..., come out of the host's
// queue in the order of their binary indices: the wide index of binary node b = the number of internal nodes at such levels before b. One flag
// pass, one exclusive scan. (The root is a wide node even when it is a leaf: collapse_wide's "a shape whose binary root is a leaf".)
struct LevelTable {
  int first[130];
  int levels;
};
__global__ void k_wide_flags(int num_nodes, const float* nodes, LevelTable lt, int L, unsigned int* flag) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > num_nodes) return;
  unsigned int f = 0;
  if (b < num_nodes) {
    int lo = 0, hi = lt.levels;  // level l holds [first[l], first[l + 1])
    while (hi - lo > 1) {
      int mid = (lo + hi) / 2;
      if (lt.first[mid] <= b) lo = mid;
      else hi = mid;
    }
    const bool internal = (__float_as_int(nodes[8 * (size_t)b + 7]) & 0x10000) != 0;
    f = (b == 0 || (internal && lo % L == 0)) ? 1u : 0u;
  }
  flag[b] = f;  // flag[num_nodes] = 0: the scan's last entry is the count
}
// One thread per wide node: its W slots {min.xyz, max.x}{max.yz, ref, axes} at blob + 32 * (node_off + W * index + slot), references ABSOLUTE
// (a child wide node's first slot in the blob; a leaf = tag | count << 27 | its first test record) as the traversal kernels read them
// (what a device pass over the host's 4-wide collapse and a host pass over its 8- / 16-wide ones made until round 5). W = 4: bits 8-11 of the
// axes word = the occupied slots (dev_lane.h).
template <int L>
__global__ void k_wide_collapse(int num_nodes, const float* nodes, const unsigned int* flag, const unsigned int* widx, int lines,
    unsigned int node_off, unsigned int test_off, float4* blob) {
  constexpr int W = 1 << L;
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= num_nodes || !flag[b]) return;
  const float inf = __int_as_float(0x7f800000);
  float4*     out = blob + 2 * ((size_t)node_off + (size_t)W * widx[b]);
  float    bmin[W][3], bmax[W][3];
  unsigned ref[W];
  for (int s = 0; s < W; s++) {
    for (int k = 0; k < 3; k++) bmin[s][k] = inf, bmax[s][k] = -inf;
    ref[s] = 0xFFFFFFFFu;
  }
  unsigned axes = 0;
  constexpr int axes_at[4] = {0, 2, 6, 14};
  // the L levels below b, depth first (the host's walk; the order of the visits does not matter here: indices come from the scan)
  int      st_node[2 * L + 2], st_level[2 * L + 2], sp = 0;
  unsigned st_path[2 * L + 2];
  st_node[0] = b, st_level[0] = 0, st_path[0] = 0u, sp = 1;
  while (sp > 0) {
    sp--;
    const int      nb = st_node[sp], level = st_level[sp];
    const unsigned path = st_path[sp];
    const float*   nd = nodes + 8 * (size_t)nb;
    const int      start = __float_as_int(nd[6]), meta = __float_as_int(nd[7]);
    const bool     internal = (meta & 0x10000) != 0;
    if (level == L || !internal) {  // becomes a slot: the first of its group
      if (level == 0 && internal) continue;  // (cannot happen: an internal root is walked)
      const int s = (int)(path << (L - level));
      for (int k = 0; k < 3; k++) bmin[s][k] = nd[k], bmax[s][k] = nd[3 + k];
      if (internal) ref[s] = node_off + (unsigned)W * widx[nb];
      else ref[s] = 0xC0000000u | ((unsigned)(meta & 0xFFFF) << 27) | (test_off + (unsigned)start * (lines ? 1u : 2u));
      continue;
    }
    axes |= (unsigned)((meta >> 24) & 3) << (axes_at[level] + 2 * (int)path);
    st_node[sp] = start + 1, st_level[sp] = level + 1, st_path[sp] = (path << 1) | 1u, sp++;
    st_node[sp] = start + 0, st_level[sp] = level + 1, st_path[sp] = (path << 1) | 0u, sp++;
  }
  unsigned word = axes;
  if (W == 4) {
    unsigned occupied = 0;
    for (int s = 0; s < W; s++) occupied |= (ref[s] != 0xFFFFFFFFu ? 1u : 0u) << s;
    word = (axes & 0xFFu) | (occupied << 8);
  }
  for (int s = 0; s < W; s++) {
    out[2 * s]     = make_float4(bmin[s][0], bmin[s][1], bmin[s][2], bmax[s][0]);
    out[2 * s + 1] = make_float4(bmax[s][1], bmax[s][2], __uint_as_float(ref[s]), __uint_as_float(word));
  }
}

// boxes: n x (min[3], max[3]) on the HOST. Outputs on the host: nodes8 (8 floats per node: box,
// start, num | internal << 16 | axis << 24 — the layout of yh_bvh_build), primitives (n ints).
// *num_nodes / *depth receive the node count and the number of levels. nodes8 must hold
// 2 * n + 1 records. Returns 0 or a hipError_t.
extern "C" int yhk_bvh_build_gpu(int n, const float* boxes, float* nodes8, int* primitives, int* num_nodes, int* depth,
    hipStream_t stream) {
  *num_nodes = 1, *depth = 1;
  if (n <= 0) {
    float* o = nodes8;
    for (int k = 0; k < 3; k++) o[k] = 3.402823466e+38f, o[3 + k] = -3.402823466e+38f;
    int z = 0;
    memcpy(o + 6, &z, 4), memcpy(o + 7, &z, 4);
    return 0;
  }
  const size_t N = (size_t)n;
  Buf d_boxes, d_nodes, d_pid;
  BVH_CHECK(d_boxes.alloc(N * 24));
  BVH_CHECK(d_nodes.alloc((2 * N + 1) * 32));
  BVH_CHECK(d_pid.alloc(N * 4));
  BVH_CHECK(hipMemcpyAsync(d_boxes.p, boxes, N * 24, hipMemcpyHostToDevice, stream));
  int level_first[130];
  int rc = build_resident(n, d_boxes.as<float>(), d_nodes.as<float>(), d_pid.as<int>(), num_nodes, depth, level_first, stream);
  if (rc) return rc;
  BVH_CHECK(hipMemcpyAsync(nodes8, d_nodes.p, (size_t)*num_nodes * 32, hipMemcpyDeviceToHost, stream));
  BVH_CHECK(hipMemcpyAsync(primitives, d_pid.p, N * 4, hipMemcpyDeviceToHost, stream));
  BVH_CHECK(hipStreamSynchronize(stream));
  return 0;
}

// ---- the device-resident pieces, for host/scene_upload.cpp (all pointers are DEVICE pointers unless said otherwise) ----
extern "C" int yhk_prim_boxes(int lines, int n, const float* pos, const float* radius, const int* idx, float* boxes, hipStream_t stream) {
  if (n > 0) hipLaunchKernelGGL(k_prim_boxes, dim3((n + 255) / 256), dim3(256), 0, stream, lines, n, pos, radius, idx, boxes);
  return (int)hipGetLastError();
}
// n >= 1. level_first: 130 ints on the host.
extern "C" int yhk_bvh_build_resident(int n, const float* d_boxes, float* d_nodes, int* d_pid, int* num_nodes, int* levels, int* level_first,
    hipStream_t stream) {
  return build_resident(n, d_boxes, d_nodes, d_pid, num_nodes, levels, level_first, stream);
}
extern "C" int yhk_leaf_records(int lines, int n, const int* pid, const float* pos, const float* nrm, const float* radius, const int* idx,
    void* prims_out, hipStream_t stream) {
  if (n > 0) hipLaunchKernelGGL(k_leaf_records, dim3((n + 255) / 256), dim3(256), 0, stream, lines, n, pid, pos, nrm, radius, idx, (float4*)prims_out);
  return (int)hipGetLastError();
}
// Wide index of every binary node for wide nodes of 2^L slots (d_flag, d_widx: num_nodes + 1 entries each); *count on the host (synchronises).
extern "C" int yhk_wide_index(int num_nodes, const float* d_nodes, int levels, const int* level_first, int L, unsigned int* d_flag,
    unsigned int* d_widx, int* count, hipStream_t stream) {
  if (levels < 1 || levels > 128 || L < 2 || L > 4) return (int)hipErrorInvalidValue;
  LevelTable lt;
  for (int l = 0; l <= levels; l++) lt.first[l] = level_first[l];
  lt.levels = levels;
  hipLaunchKernelGGL(k_wide_flags, dim3((num_nodes + 1 + 255) / 256), dim3(256), 0, stream, num_nodes, d_nodes, lt, L, d_flag);
  size_t tmp_bytes = 0;
  BVH_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_flag, d_widx, num_nodes + 1, stream));
  Buf d_tmp;
  BVH_CHECK(d_tmp.alloc(tmp_bytes));
  BVH_CHECK(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_flag, d_widx, num_nodes + 1, stream));
  unsigned int c = 0;
  BVH_CHECK(hipMemcpyAsync(&c, d_widx + num_nodes, 4, hipMemcpyDeviceToHost, stream));
  BVH_CHECK(hipStreamSynchronize(stream));
  *count = (int)c;
  return 0;
}
// The wide nodes of 2^L slots into the blob (32-byte units: node_off = the shape's first wide node, test_off = its first test record).
extern "C" int yhk_wide_collapse(int L, int num_nodes, const float* d_nodes, const unsigned int* d_flag, const unsigned int* d_widx, int lines,
    long long node_off, long long test_off, void* blob, hipStream_t stream) {
  dim3 g((num_nodes + 127) / 128), t(128);
  if (L == 2) hipLaunchKernelGGL(k_wide_collapse<2>, g, t, 0, stream, num_nodes, d_nodes, d_flag, d_widx, lines, (unsigned)node_off, (unsigned)test_off, (float4*)blob);
  else if (L == 3) hipLaunchKernelGGL(k_wide_collapse<3>, g, t, 0, stream, num_nodes, d_nodes, d_flag, d_widx, lines, (unsigned)node_off, (unsigned)test_off, (float4*)blob);
  else if (L == 4) hipLaunchKernelGGL(k_wide_collapse<4>, g, t, 0, stream, num_nodes, d_nodes, d_flag, d_widx, lines, (unsigned)node_off, (unsigned)test_off, (float4*)blob);
  else return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}
