// Developer microbenchmark (GPU box): what a packed-FP32 vector instruction (v_pk_mul_f32 / v_pk_add_f32: two results per lane) costs next to two plain ones
// (v_mul_f32 / v_sub_f32) — for ONE wave alone on its SIMD and for 2 / 4 / 8 waves sharing it. The box tests of the traversal kernels are 6 subtractions + 6
// multiplications per box on operands that a node record could hold as aligned pairs: whether packing them halves their issue time is what this answers.
// build: hipcc -O2 --offload-arch=gfx950 -o pk_f32_issue pk_f32_issue.hip      run: ./pk_f32_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void k(float* out, long long* cycles, int iters) {
  // 8 independent chains of pairs: {a_i, b_i}
  v2f   r[8];
  float s = 1.0f + 1e-7f * threadIdx.x;
  for (int i = 0; i < 8; i++) r[i] = v2f{s + i, s - i};
  v2f m = v2f{1.0000001f, 0.9999999f};
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {  // plain: 16 v_mul_f32 per round of the 8 pairs
      REP16(
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[0].x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[0].y) : "v"(m.y));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[1].x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[1].y) : "v"(m.y));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[2].x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[2].y) : "v"(m.y));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[3].x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[3].y) : "v"(m.y));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[4].x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[4].y) : "v"(m.y));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[5].x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[5].y) : "v"(m.y));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[6].x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[6].y) : "v"(m.y));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[7].x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[7].y) : "v"(m.y));
      )
    } else if (MODE == 1) {  // packed: 8 v_pk_mul_f32 for the same 16 products
      REP16(
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[0]) : "v"(m)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[1]) : "v"(m));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[2]) : "v"(m)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[3]) : "v"(m));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[4]) : "v"(m)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[5]) : "v"(m));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[6]) : "v"(m)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[7]) : "v"(m));
      )
    } else if (MODE == 2) {  // packed with operand selection: both halves of the result take the LOW half of the second source (a splat), the first source negated: (m.x - r)
      REP16(
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(r[0]) : "v"(m)); asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(r[1]) : "v"(m));
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(r[2]) : "v"(m)); asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(r[3]) : "v"(m));
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(r[4]) : "v"(m)); asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(r[5]) : "v"(m));
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(r[6]) : "v"(m)); asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "+v"(r[7]) : "v"(m));
      )
    }
  }
  long long t1 = clock64();
  float acc = 0;
  for (int i = 0; i < 8; i++) acc += r[i].x + r[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, iters = 2000;
  float*     d_out;
  long long* d_cyc;
  hipMalloc(&d_out, (size_t)cus * 2048 * 4), hipMalloc(&d_cyc, cus * 8);
  printf("%s, %d CUs; clock64 ticks (100 MHz constant clock? printed as ticks) per 16 results per lane\n", p.name, cus);
  const char* names[3] = {"2 x v_mul_f32 (16 plain)", "v_pk_mul_f32 (8 packed)", "v_pk_add_f32 op_sel/neg (8 packed)"};
  for (int wps : {1, 2, 4, 8})  // waves per SIMD: one block of 4 * wps waves per CU
    for (int mode = 0; mode < 3; mode++) {
      const int threads = 64 * 4 * wps;
      if (threads > 1024) {  // two blocks per CU instead
      }
      hipEvent_t e0, e1;
      hipEventCreate(&e0), hipEventCreate(&e1);
      auto launch = [&](int it) {
        const int blocks = threads > 1024 ? cus * (threads / 1024) : cus, th = threads > 1024 ? 1024 : threads;
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(th), 0, 0, d_out, d_cyc, it);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(th), 0, 0, d_out, d_cyc, it);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(th), 0, 0, d_out, d_cyc, it);
      };
      launch(10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch(iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      // results per lane: iters * 16 rounds * 16 results; wall ns per (16 results of one wave) per SIMD = ms * 1e6 / (iters * 16) / wps
      printf("%d wave(s) per SIMD  %-36s %8.3f ms  -> %6.2f ns per 16 results per wave on its SIMD (x 2.4 GHz = %5.1f cycles), %6.2f TFLOP/s-equivalent results\n", wps, names[mode], ms,
          ms * 1e6 / (iters * 16.0) / wps, ms * 1e6 / (iters * 16.0) / wps * 2.4, (double)cus * 4 * wps * 64 * iters * 16.0 * 16 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
