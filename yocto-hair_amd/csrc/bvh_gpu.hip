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
  const int    T  = 256;
  const size_t N  = (size_t)n;
  Buf d_boxes, d_c[2][3], d_pid[2], d_seg, d_flag, d_scan, d_fpos, d_tpos, d_segs[2], d_work, d_sflag, d_rank, d_nodes, d_tmp;
  BVH_CHECK(d_boxes.alloc(N * 24));
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
  BVH_CHECK(d_nodes.alloc((2 * N + 1) * 32));
  size_t tmp_bytes = 0;
  BVH_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_flag.as<unsigned int>(), d_scan.as<unsigned int>(), (int)(N + 1), stream));
  BVH_CHECK(d_tmp.alloc(tmp_bytes));
  BVH_CHECK(hipMemcpyAsync(d_boxes.p, boxes, N * 24, hipMemcpyHostToDevice, stream));
  int cur = 0;
  hipLaunchKernelGGL(k_centers, dim3((n + T - 1) / T), dim3(T), 0, stream, n, d_boxes.as<float>(), d_c[0][0].as<float>(),
      d_c[0][1].as<float>(), d_c[0][2].as<float>(), d_pid[0].as<int>(), d_seg.as<int>());
  Seg root{0, n, 0, 0};
  BVH_CHECK(hipMemcpyAsync(d_segs[0].p, &root, sizeof(Seg), hipMemcpyHostToDevice, stream));
  int m = 1, nodes_so_far = 1, levels = 0, segbuf = 0;
  int level_first[128];
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
        d_nodes.as<float>(), d_segs[segbuf ^ 1].as<Seg>());
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
    hipLaunchKernelGGL(k_boxes, dim3((count + T - 1) / T), dim3(T), 0, stream, first, count, d_nodes.as<float>(),
        d_boxes.as<float>(), d_pid[cur].as<int>());
  }
  BVH_CHECK(hipMemcpyAsync(nodes8, d_nodes.p, (size_t)nodes_so_far * 32, hipMemcpyDeviceToHost, stream));
  BVH_CHECK(hipMemcpyAsync(primitives, d_pid[cur].p, N * 4, hipMemcpyDeviceToHost, stream));
  BVH_CHECK(hipStreamSynchronize(stream));
  BVH_CHECK(hipGetLastError());
  *num_nodes = nodes_so_far, *depth = levels;
  return 0;
}
