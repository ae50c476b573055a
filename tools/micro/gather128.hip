// Developer micro-benchmark (GPU box): what does a wavefront pay for "every lane fetches its own random 128-byte record"
// — the memory side of one step of the one-lane traversal (csrc/dev_lane.h) — depending on HOW the record is fetched?
//
//   mode 0  eight global_load_dwordx4 per lane, all on the lane's own line (what lane_step does: every instruction
//           touches as many lines as there are busy lanes)
//   mode 1  four of them (64 bytes of the line)          mode 2  two (32 bytes)          mode 3  one (16 bytes)
//   mode 4  whole lines, cooperatively: instruction k fetches the records of lanes 8k..8k+7, eight lanes x 16 bytes
//           per line, by LDS-DMA (global_load_lds_dwordx4, per-lane source address); every lane then reads its own
//           record back from LDS (8 x ds_read_b128, XOR-swizzled so that the reads are conflict-free)
//   mode 5  as 4 with plain loads + ds_write_b128 (register staging)
//   mode 6  64-byte records, cooperatively: four lanes per record, four instructions (LDS-DMA)
//   mode 7  k_stream's mix: 5 of 8 lanes fetch 128 bytes (a wide node), 3 of 8 fetch 64 (a leaf's two segments), per-lane loads
//   mode 8  every lane 64 bytes of a 64-byte record (what 64-byte nodes would make of mode 7), per-lane loads
//   mode 9  5 of 8 lanes 64 bytes, 3 of 8 lanes 96 (64-byte nodes + leaves with a 32-byte header), per-lane loads
//
// Every lane runs a dependent chain (the next index comes out of the record), `filler` independent v_fma per step stand
// for the step's arithmetic, `active` lanes of each wave take part (the others idle, as in k_stream's trace stage).
// usage: gather128 [table MB] [steps] [filler] [active lanes] [waves per SIMD 1..4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>

#define LDS __attribute__((address_space(3)))
typedef float v4f __attribute__((ext_vector_type(4)));

#ifndef MINW
#define MINW 4
#endif
template <int MODE>
__global__ __launch_bounds__(256, MINW) void gather(const v4f* __restrict__ recs, unsigned mask, int steps, int filler, int active, float* sink) {
  extern __shared__ v4f lds[];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  LDS v4f* stage = (LDS v4f*)lds + wib * 512;  // 8 KB per wave: 64 records x 8 chunks
  unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  idx &= mask;
  float acc0 = 0, acc1 = 1, acc2 = 2, acc3 = 3;
  const bool on = lane < active;
  for (int s = 0; s < steps; s++) {
    v4f r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = v4f{0, 0, 0, 0};
    if (MODE >= 7) {
      if (on) {
        const bool    nodey = (lane & 7) < 5;
        const v4f*    a = MODE == 7 ? recs + (size_t)idx * 8 : recs + (size_t)(idx * 2u + (lane & 1)) * 4;  // (modes 8, 9: 64-byte records, twice as many)
        r[0] = a[0], r[1] = a[1], r[2] = a[2], r[3] = a[3];
        if (MODE == 7 && nodey) r[4] = a[4], r[5] = a[5], r[6] = a[6], r[7] = a[7];
        if (MODE == 9 && !nodey) r[4] = a[4], r[5] = a[5];
      }
    } else if (MODE <= 3) {
      if (on) {
        const v4f*    a = recs + (size_t)idx * 8;
        constexpr int N = MODE == 0 ? 8 : MODE == 1 ? 4 : MODE == 2 ? 2 : 1;
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = a[k];
      }
    } else if (MODE == 4 || MODE == 5) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int      src  = 8 * k + (lane >> 3);                // the lane whose record this instruction's octet fetches
        const unsigned sidx = (unsigned)__shfl((int)idx, src);
        const int      c    = (lane & 7) ^ ((src >> 1) & 7);      // swizzle on the SOURCE side: LDS position p holds chunk p ^ f(record)
        const v4f*     g    = recs + (size_t)sidx * 8 + c;
        if (src < active) {
          if (MODE == 4) __builtin_amdgcn_global_load_lds((const void*)g, (LDS void*)(stage + 64 * k), 16, 0, 0);
          else stage[64 * k + lane] = *g;
        }
      }
      if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (on) {
#pragma unroll
        for (int c = 0; c < 8; c++) r[c] = stage[lane * 8 + (c ^ ((lane >> 1) & 7))];
      }
    } else {  // MODE 6: 64-byte records, four lanes per record
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int      src  = 16 * k + (lane >> 2);
        const unsigned sidx = (unsigned)__shfl((int)idx, src);
        const int      c    = (lane & 3) ^ ((src >> 2) & 3);
        const v4f*     g    = recs + (size_t)sidx * 4 + c;
        if (src < active) __builtin_amdgcn_global_load_lds((const void*)g, (LDS void*)(stage + 64 * k), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (on) {
#pragma unroll
        for (int c = 0; c < 4; c++) r[c] = stage[lane * 4 + (c ^ ((lane >> 2) & 3))];
      }
    }
    if (on) {
      float x = 0;
#pragma unroll
      for (int k = 1; k < 8; k++) x += r[k].x + r[k].y + r[k].z + r[k].w;
      x += r[0].y + r[0].z + r[0].w;
      acc0 += x;
      idx = __float_as_uint(r[0].x) & (MODE == 6 ? 2 * mask + 1 : mask);  // next record: a dependent chain
      for (int f = 0; f < filler; f += 4) {
        acc0 = __builtin_fmaf(acc0, 1.0001f, 0.5f), acc1 = __builtin_fmaf(acc1, 1.0001f, 0.5f);
        acc2 = __builtin_fmaf(acc2, 1.0001f, 0.5f), acc3 = __builtin_fmaf(acc3, 1.0001f, 0.5f);
      }
    }
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + (float)idx;
}

typedef void (*kern_t)(const v4f*, unsigned, int, int, int, float*);

int main(int argc, char** argv) {
  size_t mb      = argc > 1 ? atoi(argv[1]) : 256;
  int    steps   = argc > 2 ? atoi(argv[2]) : 1000;
  int    filler  = argc > 3 ? atoi(argv[3]) : 600;
  int    active  = argc > 4 ? atoi(argv[4]) : 44;
  int    wps     = argc > 5 ? atoi(argv[5]) : 4;
  size_t nrec    = mb * 1024 * 1024 / 128;  // power of two
  unsigned mask  = (unsigned)nrec - 1;
  std::vector<float> h(nrec * 32);
  std::mt19937 rng(7);
  for (size_t i = 0; i < nrec * 32; i++) h[i] = 1e-3f * (float)(rng() & 1023);
  for (size_t i = 0; i < nrec; i++) {  // the chain's index sits in the first dword of chunk 0 (and of chunk 4: 64-byte records)
    unsigned nx = rng();
    memcpy(&h[i * 32], &nx, 4);
    nx = rng();
    memcpy(&h[i * 32 + 16], &nx, 4);
  }
  v4f* d; float* sink;
  hipMalloc(&d, nrec * 128); hipMalloc(&sink, 1024 * 256 * 4);
  hipMemcpy(d, h.data(), nrec * 128, hipMemcpyHostToDevice);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  int blocks = prop.multiProcessorCount * wps;  // 256-thread blocks: wps waves per SIMD
  kern_t ks[10] = {gather<0>, gather<1>, gather<2>, gather<3>, gather<4>, gather<5>, gather<6>, gather<7>, gather<8>, gather<9>};
  const char* names[10] = {"8 x dwordx4 per lane", "4 x dwordx4 per lane", "2 x dwordx4 per lane", "1 x dwordx4 per lane", "whole lines by LDS-DMA", "whole lines, register staged", "64-B records by LDS-DMA", "mix: 5/8 lanes 128 B, 3/8 64 B", "every lane 64 B of a 64-B record", "mix: 5/8 lanes 64 B, 3/8 96 B"};
  printf("# table %zu MB, %d steps, filler %d fma, %d active lanes, %d waves per SIMD (%d blocks)\n", mb, steps, filler, active, wps, blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int m = 0; m < 10; m++) {
    size_t ldsb = (m >= 4 && m <= 6) ? 32768 : 0;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(256), ldsb, 0, d, mask, steps, filler, active, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    double wave_steps = (double)blocks * 4 * steps;
    printf("mode %d %-30s %8.3f ms  %7.1f ns per wave step per SIMD  %6.1f G lane-steps/s  %5.2f TB/s of 128-B lines %s\n", m, names[m], best,
           best * 1e6 / (wave_steps / (prop.multiProcessorCount * 4)), wave_steps * active / best / 1e6, wave_steps * active * 128 / best / 1e9,
           e == hipSuccess ? "" : hipGetErrorString(e));
  }
  return 0;
}
