// kernels.hip — gfx950 kernels of the hair path and their launchers.
//
//   k_trace          the sample loop (trace_samples, pt.cpp:1992-2007) with a QUAD per path: persistent
//                    wavefronts pull 4x4-pixel work items (most expensive first); four adjacent lanes own one
//                    pixel / one path (its PCG32 stream is sequential) and split BVH steps and hair lobes over
//                    their lanes. In LDS: the traversal stacks (one column per quad, sized by the scene's
//                    need) and the tables of dev_trace.h: stage_tables — scene level, camera, small area
//                    lights, the index of the environment cdf, the material table. Two launch shapes: 512 threads
//                    x 4 waves per SIMD, 256 x 5. The other sample-loop kernels: csrc/wide.hip (more lanes per
//                    path), csrc/stream.hip (k_stream, one lane per path); the host picks per launch by measurement.
//   k_hair_*         unit-level batches of the four yocto::extension functions
//   k_intersect      unit-level closest-hit batch
//   k_selftest       the four Monte-Carlo self-tests (ext.cpp:555-693), made
//                    data-parallel with PCG32 jump-ahead (draw counts per
//                    iteration are fixed, so sample i's stream offset is known)
//   k_pack / k_unpack  tile-packed float4 framebuffer for the RCCL gather
//
// Compiled with -ffp-contract=off (see dev_math.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "yhair.h"
#include "dev_items.h"

template <int SHADER>
__global__ __launch_bounds__(YH_BLOCK, YH_MIN_WAVES) void k_trace_shader(const yhd_scene sc, const yhd_state st,
    int nsamples, yhd_counters* counters) {
  trace_items<false, true, YH_BLOCK, SHADER>(sc, st, nsamples, counters);
}

// render[ij] = accumulated / samples (pt.cpp:1688) into a full W*H image
__global__ void k_resolve(const yhd_state st, int samples, yhd_float4* image) {
  // block k handles the k-th tile owned by this shard: tile id rank + k * world
  int lane = threadIdx.x;
  int tile = st.shard_rank + blockIdx.x * st.shard_world;
  int i    = (tile % st.tiles_x) * YH_TILE + (lane & 7);
  int j    = (tile / st.tiles_x) * YH_TILE + (lane >> 3);
  if (i >= st.width || j >= st.height) return;
  size_t     pix = (size_t)j * st.width + i;
  yhd_float4 a   = st.accum[pix];
  float      n   = (float)samples;
  image[pix]     = samples > 0 ? yhd_float4{a.x / n, a.y / n, a.z / n, a.w / n} : yhd_float4{0, 0, 0, 0};
}
// tile-packed variant: 64 float4 per owned tile, tiles in increasing id order
__global__ void k_pack(const yhd_state st, int samples, yhd_float4* packed) {
  int lane = threadIdx.x;
  int tile = st.shard_rank + blockIdx.x * st.shard_world;
  int i    = (tile % st.tiles_x) * YH_TILE + (lane & 7);
  int j    = (tile / st.tiles_x) * YH_TILE + (lane >> 3);
  yhd_float4 out = {0, 0, 0, 0};
  if (i < st.width && j < st.height && samples > 0) {
    yhd_float4 a = st.accum[(size_t)j * st.width + i];
    float      n = (float)samples;
    out          = yhd_float4{a.x / n, a.y / n, a.z / n, a.w / n};
  }
  packed[(size_t)blockIdx.x * 64 + lane] = out;
}
__global__ void k_unpack(const yhd_float4* packed, int src_rank, int world, int num_tiles_total,
    int tiles_x, int width, int height, yhd_float4* image) {
  // the k-th tile owned by src_rank is tile id src_rank + k * world
  int k = blockIdx.x, lane = threadIdx.x;
  int tile = src_rank + k * world;
  if (tile >= num_tiles_total) return;
  int i = (tile % tiles_x) * YH_TILE + (lane & 7);
  int j = (tile / tiles_x) * YH_TILE + (lane >> 3);
  if (i < width && j < height) image[(size_t)j * width + i] = packed[(size_t)k * 64 + lane];
}

// ---------------------------------------------------------------------------
// Unit-level batches
// ---------------------------------------------------------------------------
// hair_brdf as 30 floats (yhair.h): sigma_a[3] alpha eta h v[4] s sin[3]
// cos[3] gamma_o world_to_brdf[12]
YH_DEV void unpack_brdf(const float* b, yhd_material& m, hair_hit& hh) {
  m.sigma_a[0] = b[0], m.sigma_a[1] = b[1], m.sigma_a[2] = b[2];
  m.alpha = b[3], m.eta = b[4];
  hh.h = b[5];
  for (int p = 0; p < 4; p++) m.v[p] = b[6 + p];
  m.s = b[10];
  for (int k = 0; k < 3; k++) m.sin_2k_alpha[k] = b[11 + k], m.cos_2k_alpha[k] = b[14 + k];
  hh.gamma_o = b[17];
  hh.w2b     = ldframe(b + 18);
  derive_material(m);
}

// eval_hair_brdf (ext.cpp:127-177) entirely on the device
#include "yhair.h"
typedef yh_material yh_material_in;  // the public struct itself (include/yhair.h)
template <int N>
YH_DEV float powt(float v) {  // ext.cpp:95-109
  if constexpr (N == 0) return 1;
  else if constexpr (N == 1) return v;
  else {
    float n2 = powt<N / 2>(v);
    return n2 * n2 * powt<(N & 1)>(v);
  }
}
__global__ void k_hair_brdf(int n, const yh_material_in* mats, const float* v, const float* nrm,
    const float* tng, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const yh_material_in& m = mats[i];
  f3 sigma_a = mk3(0.0f);
  f3 msa = ld3(m.sigma_a), col = ld3(m.color);
  if (!is_zero(msa)) {
    sigma_a = msa;
  } else if (!is_zero(col)) {  // sigma_a_from_reflectance (ext.cpp:121-125)
    float bn  = m.beta_n;
    float den = 5.969f - 0.215f * bn + 2.532f * sqr(bn) - 10.73f * powt<3>(bn) +
                5.574f * powt<4>(bn) + 0.245f * powt<5>(bn);
    f3 q    = f3{logf(col.x), logf(col.y), logf(col.z)} / den;
    sigma_a = q * q;
  } else if (m.eumelanin != 0 || m.pheomelanin != 0) {  // ext.cpp:115-119
    sigma_a = m.eumelanin * f3{0.419f, 0.697f, 1.37f} + m.pheomelanin * f3{0.187f, 0.4f, 1.05f};
  }
  float bm = m.beta_m, bn = m.beta_n;
  float* o = out + 30 * (size_t)i;
  o[0] = sigma_a.x, o[1] = sigma_a.y, o[2] = sigma_a.z;
  o[3] = m.alpha, o[4] = m.eta;
  hair_hit hh = hair_setup(v[i], ld3(nrm + 3 * i), ld3(tng + 3 * i));
  o[5]     = hh.h;
  float v0 = sqr(0.726f * bm + 0.812f * sqr(bm) + 3.7f * powt<20>(bm));
  o[6] = v0, o[7] = 0.25f * v0, o[8] = 4 * v0, o[9] = 4 * v0;
  o[10] = 0.626657069f * (0.265f * bn + 1.194f * sqr(bn) + 5.372f * powt<22>(bn));
  float s0 = sinf(pif / 180 * m.alpha);
  float c0 = exact_safe_sqrt(1 - sqr(s0));
  float s1 = 2 * c0 * s0, c1 = sqr(c0) - sqr(s0);
  float s2 = 2 * c1 * s1, c2 = sqr(c1) - sqr(s1);
  o[11] = s0, o[12] = s1, o[13] = s2, o[14] = c0, o[15] = c1, o[16] = c2;
  o[17] = hh.gamma_o;
  const frame& w = hh.w2b;
  o[18] = w.x.x, o[19] = w.x.y, o[20] = w.x.z, o[21] = w.y.x, o[22] = w.y.y, o[23] = w.y.z;
  o[24] = w.z.x, o[25] = w.z.y, o[26] = w.z.z, o[27] = w.o.x, o[28] = w.o.y, o[29] = w.o.z;
}
// eval and pdf run the INTEGRATOR's code path: one quad (four threads) per item,
// lobe p on lane p (hair_eval_pdf_quad). The scalar hair_eval_pdf<> is what the
// self-test kernel uses, so both formulations are checked against the reference.
__global__ void k_hair_eval(int n, const float* brdf, const float* wo, const float* wi, float* out) {
  int  i     = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  bool valid = i < n;
  if (!valid) i = n - 1;  // keep whole quads converged
  yhd_material m;
  hair_hit     hh;
  unpack_brdf(brdf + 30 * (size_t)i, m, hh);
  f3    f;
  float pdf;
  hair_eval_pdf_quad(m, hh, ld3(wo + 3 * i), ld3(wi + 3 * i), f, pdf);
  if (valid && (threadIdx.x & 3) == 0) out[3 * i] = f.x, out[3 * i + 1] = f.y, out[3 * i + 2] = f.z;
}
__global__ void k_hair_pdf(int n, const float* brdf, const float* wo, const float* wi, float* out) {
  int  i     = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  bool valid = i < n;
  if (!valid) i = n - 1;
  yhd_material m;
  hair_hit     hh;
  unpack_brdf(brdf + 30 * (size_t)i, m, hh);
  f3    f;
  float pdf;
  hair_eval_pdf_quad(m, hh, ld3(wo + 3 * i), ld3(wi + 3 * i), f, pdf);
  if (valid && (threadIdx.x & 3) == 0) out[i] = pdf;
}
__global__ void k_hair_sample(int n, const float* brdf, const float* wo, const float* rn, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  yhd_material m;
  hair_hit     hh;
  unpack_brdf(brdf + 30 * (size_t)i, m, hh);
  f3 w = hair_sample(m, hh, ld3(wo + 3 * i), rn[2 * i], rn[2 * i + 1]);
  out[3 * i] = w.x, out[3 * i + 1] = w.y, out[3 * i + 2] = w.z;
}
// four threads (one quad) per ray; block of 256 threads = 64 quads
__global__ __launch_bounds__(256) void k_intersect(const yhd_scene sc, int n, const float* rays, int* object,
    int* element, float* uv, float* dist) {
  __shared__ unsigned int stacks[(YH_QSTACK + YH_HITROWS) * 64];
  int  i     = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  bool valid = i < n;
  if (!valid) i = n - 1;  // whole quads stay converged; surplus quads redo the last ray
  trace_ctx tc;
  tc.sc = &sc, tc.stats = nullptr, tc.lds_scene = nullptr, tc.ls = nullptr, tc.sc_dev = nullptr, tc.lds_lights = nullptr, tc.lds_envtab = nullptr, tc.lds_mats = nullptr;
  tc.lds_stack   = (YH_LDS unsigned int*)stacks + (threadIdx.x >> 2);
  const float* r = rays + 8 * (size_t)i;
  ray_t ray      = ray_t{ld3(r), ld3(r + 3), r[6], r[7]};
  hit_t h        = trace_ray<false, 64>(tc, ray, -1);
  if (valid && (threadIdx.x & 3) == 0) {
    object[i] = h.object, element[i] = hit_element(sc, h);
    uv[2 * i] = h.u, uv[2 * i + 1] = h.v, dist[i] = h.distance;
  }
}

// ---------------------------------------------------------------------------
// Self-tests (ext.cpp:555-693)
// ---------------------------------------------------------------------------
// PCG32 jump-ahead: state after `delta` steps of the LCG (Brown, "Random
// number generation with arbitrary strides").
YH_DEV uint64_t pcg_advance(uint64_t state, uint64_t inc, uint64_t delta) {
  uint64_t cur_mult = 6364136223846793005ULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
  while (delta > 0) {
    if (delta & 1) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta /= 2;
  }
  return acc_mult * state + acc_plus;
}
YH_DEV f3 sample_sphere(float rx, float ry) {  // math.h:4847-4852
  float z   = 2 * ry - 1;
  float r   = sqrtf(fclamp(1 - z * z, 0.0f, 1.0f));
  float phi = 2 * pif * rx;
  return f3{r * cosf(phi), r * sinf(phi), z};
}
struct selftest_args {
  int      which;
  float    beta_m, beta_n;
  uint64_t state, inc;   // rng state at the start of this (beta_m, beta_n) block's sample loop
  int      count;
  float    wo[3];        // tests 0, 1, 3
};
// out (double): [0..2] first sum, [3..5] second sum, [6] max |lum(f)/pdf - 1| as float bits via atomicMax
__global__ void k_selftest(selftest_args a, double* sums, unsigned int* worst_bits) {
  int   i = blockIdx.x * blockDim.x + threadIdx.x;
  float s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
  float dev = 0.0f;
  if (i < a.count) {
    int   per_iter = a.which == 2 ? 5 : 3;
    rng_t rng;
    rng.inc   = a.inc;
    rng.state = pcg_advance(a.state, a.inc, (uint64_t)per_iter * (uint64_t)i);
    float h   = rand1f(rng);
    if (a.which == 0 && h == 0) h += 1.1920929e-07f;  // flt_eps
    // eval_hair_brdf(mat{beta_m, beta_n, alpha 0}, h, {0,0,1}, {1,0,0})
    yhd_material m;
    m.sigma_a[0] = m.sigma_a[1] = m.sigma_a[2] = 0;
    m.alpha = 0, m.eta = 1.55f;
    float bm = a.beta_m, bn = a.beta_n;
    m.v[0] = sqr(0.726f * bm + 0.812f * sqr(bm) + 3.7f * powt<20>(bm));
    m.v[1] = 0.25f * m.v[0], m.v[2] = 4 * m.v[0], m.v[3] = m.v[2];
    m.s    = 0.626657069f * (0.265f * bn + 1.194f * sqr(bn) + 5.372f * powt<22>(bn));
    float s_0 = sinf(pif / 180 * m.alpha), c_0 = exact_safe_sqrt(1 - sqr(s_0));
    m.sin_2k_alpha[0] = s_0, m.cos_2k_alpha[0] = c_0;
    for (int k = 1; k < 3; k++) {
      m.sin_2k_alpha[k] = 2 * m.cos_2k_alpha[k - 1] * m.sin_2k_alpha[k - 1];
      m.cos_2k_alpha[k] = sqr(m.cos_2k_alpha[k - 1]) - sqr(m.sin_2k_alpha[k - 1]);
    }
    derive_material(m);
    hair_hit hh = hair_setup(h, f3{0, 0, 1}, f3{1, 0, 0});
    f3 wo = ld3(a.wo);
    if (a.which == 2) {
      float wx = rand1f(rng), wy = rand1f(rng);
      wo = sample_sphere(wx, wy);
    }
    float rx = rand1f(rng), ry = rand1f(rng);
    f3    f;
    float pdf;
    if (a.which == 0) {
      f3 wi = sample_sphere(rx, ry);
      hair_eval_pdf<true, false>(m, hh, wo, wi, f, pdf);
      s0[0] = f.x, s0[1] = f.y, s0[2] = f.z;
    } else {
      f3 wi = hair_sample(m, hh, wo, rx, ry);
      hair_eval_pdf<true, true>(m, hh, wo, wi, f, pdf);
      if (a.which == 1) {
        if (pdf > 0) s0[0] = f.x / pdf, s0[1] = f.y / pdf, s0[2] = f.z / pdf;
      } else if (a.which == 2) {
        if (pdf > 0) dev = fabs_(luminance(f) / pdf - 1);
      } else {
        float li = wi.z * wi.z;  // Li(w) = w.z^2 (ext.cpp:662)
        if (pdf > 0) s0[0] = f.x * li / pdf, s0[1] = f.y * li / pdf, s0[2] = f.z * li / pdf;
        f3 wu = sample_sphere(rx, ry);
        f3 fu;
        hair_eval_pdf<true, false>(m, hh, wo, wu, fu, pdf);
        float lu = wu.z * wu.z;
        s1[0] = fu.x * lu, s1[1] = fu.y * lu, s1[2] = fu.z * lu;
      }
    }
  }
  // wave reduction, then one atomic per wave (sums are tiny: double atomics)
  for (int k = 0; k < 3; k++) {
    double a0 = s0[k], a1 = s1[k];
    for (int off = 32; off > 0; off >>= 1) {
      a0 += __shfl_down(a0, off, 64);
      a1 += __shfl_down(a1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&sums[k], a0);
      atomicAdd(&sums[3 + k], a1);
    }
  }
  for (int off = 32; off > 0; off >>= 1) dev = fmaxf(dev, __shfl_down(dev, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(worst_bits, __float_as_uint(dev));
}

// ---------------------------------------------------------------------------
// Launchers (called from the g++-compiled host code)
// ---------------------------------------------------------------------------
// One surface lobe per launch (include/yhair.h YH_LOBE_*): the unit-level
// counterpart of yocto_math.h:4427-4755 for the parity tests.
__global__ void k_surface_lobe(int kind, int n, const float* params, const float* normal, const float* wo_,
    const float* wi_, const float* rn, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* q = params + 8 * (size_t)i;
  float ior = q[0], rough = q[1];
  f3    eta = ld3(q + 2), etak = ld3(q + 5);
  f3    nn = ld3(normal + 3 * (size_t)i), wo = ld3(wo_ + 3 * (size_t)i), wi = ld3(wi_ + 3 * (size_t)i);
  float rnl = rn[3 * (size_t)i], rx = rn[3 * (size_t)i + 1], ry = rn[3 * (size_t)i + 2];
  f3    f = mk3(0.0f), w = mk3(0.0f);
  float pdf = 0;
  switch (kind) {
    case 0:
      f = eval_diffuse_reflection(nn, wo, wi), pdf = sample_diffuse_reflection_pdf(nn, wo, wi);
      w = sample_diffuse_reflection(nn, wo, rx, ry);
      break;
    case 1:
      f = eval_microfacet_reflection(ior, rough, nn, wo, wi), pdf = sample_microfacet_reflection_pdf(rough, nn, wo, wi);
      w = sample_microfacet_reflection(rough, nn, wo, rx, ry);
      break;
    case 2:
      f   = eval_microfacet_reflection(eta, etak, rough, nn, wo, wi);
      pdf = sample_microfacet_reflection_pdf(rough, nn, wo, wi);
      w   = sample_microfacet_reflection(rough, nn, wo, rx, ry);
      break;
    case 3:
      f = eval_microfacet_transmission(rough, nn, wo, wi), pdf = sample_microfacet_transmission_pdf(rough, nn, wo, wi);
      w = sample_microfacet_transmission(rough, nn, wo, rx, ry);
      break;
    case 4:
      f   = eval_microfacet_refraction(ior, rough, nn, wo, wi);
      pdf = sample_microfacet_refraction_pdf(ior, rough, nn, wo, wi);
      w   = sample_microfacet_refraction(ior, rough, nn, wo, rnl, rx, ry);
      break;
    case 5:
      f = eval_delta_reflection(ior, nn, wo, wi), pdf = sample_delta_reflection_pdf(nn, wo, wi);
      w = sample_delta_reflection(nn, wo);
      break;
    case 6:
      f = eval_delta_reflection(eta, etak, nn, wo, wi), pdf = sample_delta_reflection_pdf(nn, wo, wi);
      w = sample_delta_reflection(nn, wo);
      break;
    case 7:
      f = eval_delta_transmission(nn, wo, wi), pdf = sample_delta_transmission_pdf(nn, wo, wi);
      w = sample_delta_transmission(nn, wo);
      break;
    case 8:
      f = eval_delta_refraction(ior, nn, wo, wi), pdf = sample_delta_refraction_pdf(ior, nn, wo, wi);
      w = sample_delta_refraction(ior, nn, wo, rnl);
      break;
    default: break;
  }
  float* o = out + 7 * (size_t)i;
  o[0] = f.x, o[1] = f.y, o[2] = f.z, o[3] = pdf, o[4] = w.x, o[5] = w.y, o[6] = w.z;
}
// eval_brdf + lobe dispatch (pt.cpp:405-471, 1069-1280) of non-hair materials;
// 29 floats per item (YH_SURFACE_BSDF_FLOATS in include/yhair.h)
__global__ void k_surface_bsdf(int n, const yhd_material* mats, const float* normal, const float* wo_,
    const float* wi_, const float* rn, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f3    nn = ld3(normal + 3 * (size_t)i), wo = ld3(wo_ + 3 * (size_t)i), wi = ld3(wi_ + 3 * (size_t)i);
  float rnl = rn[3 * (size_t)i], rx = rn[3 * (size_t)i + 1], ry = rn[3 * (size_t)i + 2];
  surface_brdf_t b = surface_brdf(mats[i], nn, wo);
  float* o = out + 29 * (size_t)i;
  f3 lobes[5] = {b.diffuse, b.specular, b.metal, b.transmission, b.refraction};
  for (int k = 0; k < 5; k++) o[3 * k] = lobes[k].x, o[3 * k + 1] = lobes[k].y, o[3 * k + 2] = lobes[k].z;
  o[15] = b.roughness, o[16] = b.opacity;
  o[17] = b.diffuse_pdf, o[18] = b.specular_pdf, o[19] = b.metal_pdf, o[20] = b.transmission_pdf;
  o[21] = b.refraction_pdf;
  f3    f, w;
  float pdf;
  if (!is_delta(b)) {
    surface_eval_pdf(b, nn, wo, wi, f, pdf);
    w = surface_sample(b, nn, wo, rnl, rx, ry);
  } else {
    surface_eval_pdf_delta(b, nn, wo, wi, f, pdf);
    w = surface_sample_delta(b, nn, wo, rnl);
  }
  o[22] = f.x, o[23] = f.y, o[24] = f.z, o[25] = pdf, o[26] = w.x, o[27] = w.y, o[28] = w.z;
}

// pbrt "curve" -> five-vertex line strand (yocto_pbrt.h:1751-1797, number_sub = 4).
// One lane per output VERTEX (5 per curve): the control points are read through
// the cache by the five lanes of a curve, the 12 + 12 + 4 B vertex records and the
// 8 B line records are written once. HBM-streaming: 56 B in, 172 B out per curve.
__global__ void k_curves_to_lines(int n, const float* P, const float* width0, const float* width1, int base_vertex,
    float* positions, float* normals, float* radius, int* lines) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 5ll * n) return;
  int c = (int)(t / 5), i = (int)(t % 5);
  const float* q = P + 12 * (size_t)c;
  f3 p0 = ld3(q), p1 = ld3(q + 3), p2 = ld3(q + 6), p3 = ld3(q + 9);
  f3    pos, tan;
  float rad;
  if (i == 0) {
    pos = p0, tan = normalize(p1 - p0), rad = width0[c];
  } else if (i == 4) {
    pos = p3, tan = normalize(p3 - p2), rad = width1[c];
  } else {
    float u = (float)i / 4;
    // interpolate_bezier / _derivative (math.h:3346-3357), lerp (math.h:1803)
    pos = p0 * (1 - u) * (1 - u) * (1 - u) + p1 * 3 * u * (1 - u) * (1 - u) + p2 * 3 * u * u * (1 - u) + p3 * u * u * u;
    tan = normalize((p1 - p0) * 3 * (1 - u) * (1 - u) + (p2 - p1) * 6 * u * (1 - u) + (p3 - p2) * 3 * u * u);
    rad = width0[c] * (1 - u) + width1[c] * u;
  }
  size_t v = (size_t)t;
  positions[3 * v] = pos.x, positions[3 * v + 1] = pos.y, positions[3 * v + 2] = pos.z;
  normals[3 * v] = tan.x, normals[3 * v + 1] = tan.y, normals[3 * v + 2] = tan.z;
  radius[v] = rad;
  if (i < 4) lines[8 * (size_t)c + 2 * i] = base_vertex + (int)t, lines[8 * (size_t)c + 2 * i + 1] = base_vertex + (int)t + 1;
}

extern "C" {

trace_kernel_t yhk_wide_kernel(int counted, int general, int shape);  // csrc/wide.hip: launch shapes 4, 6, 7, 8
#define YH_DENSE_WAVES 5 /* 96 VGPRs: at 6 (80 VGPRs) the traversal loop itself spills and the kernel is at the mercy of the register allocator (measured 0.6-0.75x after an unrelated change of the shading code, profiles/r02) */
// shape 0 = 512 threads x 4 waves per SIMD, shape 1 = 256 threads x YH_DENSE_WAVES (5) waves per SIMD: quads over 4-wide nodes;
// (shape 2 was the quad form over 8-wide nodes, a closed experiment: never chosen, not built;) shape 4 = 256 x 4, octets over 8-wide nodes (YH_MODE_OCT);
// shape 6 = 256 x 4, sixteen lanes per path over 16-wide nodes (YH_MODE_HEX); shape 7 = shape 4 with leaf pairs (YH_MODE_OCTP). (3 is k_stream, csrc/stream.hip; 5 the host's
// side-by-side launch of shapes 0 and 4.) shape 8 = shape 6 with leaf groups (YH_MODE_HEXP).
static bool shape_oct(int shape) { return shape == 4 || shape == 7; }
static bool shape_hex(int shape) { return shape == 6 || shape == 8; }
static int shape_block(int shape) { return shape == 1 ? 256 : (shape_oct(shape) || shape_hex(shape)) ? YH_OCT_BLOCK : YH_BLOCK; }
static int shape_groups(int shape) { return shape_block(shape) / (shape_hex(shape) ? 16 : shape_oct(shape) ? 8 : 4); }
static trace_kernel_t trace_kernel(bool counted, bool general, int shape, int shader = YH_SHADER_PATH) {
  if (shader == YH_SHADER_NAIVE) return k_trace_shader<YH_SHADER_NAIVE>;
  if (shader == YH_SHADER_EYELIGHT) return k_trace_shader<YH_SHADER_EYELIGHT>;
  if (shader == YH_SHADER_NORMAL) return k_trace_shader<YH_SHADER_NORMAL>;
  // (instrumented builds of the 8-wide forms: plain scenes only; their per-quad counters count an octet twice, the wave-level ones hold)
  if (shape_oct(shape) || shape_hex(shape)) return yhk_wide_kernel(counted ? 1 : 0, general ? 1 : 0, shape);  // csrc/wide.hip
  if (shape == 2) return nullptr;
  // The GENERAL variants carry the surface lobes, volumes, textures and the through-memory light code. The dense shape
  // spilled 184 registers at the plain variant's 96 (7 scratch instructions inside its traversal loops): it runs at 256 x 4
  // (128 registers, 49 spilled, none in the traversal loops; lobes / volumes +15-20 %, profiles/r03/general_waves_ab.txt).
  // The 512-thread shape stays at 4 waves per SIMD (68 spilled, none in the traversal loops): at 3 (168 registers, 2
  // spilled) its expensive items no longer fit the resident waves and it loses a third.
  if (general && !counted && shape == 1) return k_trace<false, true, 256, YH_DENSE_WAVES - 1>;
  if (shape == 1)
    return counted ? (general ? k_trace<true, true, 256, YH_DENSE_WAVES> : k_trace<true, false, 256, YH_DENSE_WAVES>)
                   : (general ? k_trace<false, true, 256, YH_DENSE_WAVES> : k_trace<false, false, 256, YH_DENSE_WAVES>);
  return counted ? (general ? k_trace<true, true, YH_BLOCK, YH_MIN_WAVES> : k_trace<true, false, YH_BLOCK, YH_MIN_WAVES>)
                 : (general ? k_trace<false, true, YH_BLOCK, YH_MIN_WAVES> : k_trace<false, false, YH_BLOCK, YH_MIN_WAVES>);
}
static size_t trace_lds(const yhd_scene* sc, int shape) {
  const int entries = shape_hex(shape) ? sc->stack_entries16 : shape_oct(shape) ? sc->stack_entries8 : sc->stack_entries;
  return (size_t)(entries + YH_HITROWS) * shape_groups(shape) * 4 + (size_t)YHD_LDS_TABLES_F4(sc) * 16;
}
// `shape`: 0, 1, 4, 6, 7 or 8 (above); the caller built the work list for it (shape 4: half-quadrant entries)
int yhk_trace(const yhd_scene* sc, const yhd_state* st, int nsamples, yhd_counters* counters, int shape,
    int grid_blocks, hipStream_t stream) {
  const bool path  = st->shader == YH_SHADER_PATH;
  if (!path) shape = 0;  // the other shaders have one shape (512 x 4)
  if (!path && counters) return (int)hipErrorInvalidValue;
  if (shape == 3 || shape == 5 || shape < 0 || shape > 8) return (int)hipErrorInvalidValue;
  size_t    lds   = trace_lds(sc, shape);
  trace_kernel_t k     = trace_kernel(counters != nullptr, sc->general_materials != 0, shape, st->shader);
  if (!k) return (int)hipErrorInvalidValue;
  if (lds > 64 * 1024) {  // above 64 KB the dynamic-LDS limit must be raised explicitly; the attribute is per
                          // device, so it is set for the current device at every such launch (no process-wide cache)
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(grid_blocks), dim3(shape_block(shape)), lds, stream, *sc, *st, nsamples, counters);
  return (int)hipGetLastError();
}
int yhk_block_threads(int shape) { return shape_block(shape); }
int yhk_stack_entries(void) { return YH_QSTACK; }
int yhk_trace_lds_bytes(const yhd_scene* sc, int shape) { return (int)trace_lds(sc, shape); }
int yhk_trace_occupancy(int lds_bytes, int general, int shape) {
  int            blocks = 0;
  trace_kernel_t k      = trace_kernel(false, general != 0, shape);
  if (!k) return 0;  // a launch shape this build does not contain
  if (lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return 0;  // the kernel cannot be launched with this much LDS: the caller reports it
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k, shape_block(shape), lds_bytes) != hipSuccess) return 1;
  return blocks < 1 ? 0 : blocks;
}
int yhk_resolve(const yhd_state* st, int owned_tiles, int samples, void* image, hipStream_t stream) {
  if (owned_tiles) hipLaunchKernelGGL(k_resolve, dim3(owned_tiles), dim3(64), 0, stream, *st, samples, (yhd_float4*)image);
  return (int)hipGetLastError();
}
int yhk_pack(const yhd_state* st, int owned_tiles, int samples, void* packed, hipStream_t stream) {
  if (owned_tiles) hipLaunchKernelGGL(k_pack, dim3(owned_tiles), dim3(64), 0, stream, *st, samples, (yhd_float4*)packed);
  return (int)hipGetLastError();
}
int yhk_unpack(const void* packed, int src_rank, int world, int ntiles_src, int num_tiles_total, int tiles_x,
    int width, int height, void* image, hipStream_t stream) {
  if (ntiles_src)
    hipLaunchKernelGGL(k_unpack, dim3(ntiles_src), dim3(64), 0, stream, (const yhd_float4*)packed, src_rank, world,
        num_tiles_total, tiles_x, width, height, (yhd_float4*)image);
  return (int)hipGetLastError();
}
int yhk_curves_to_lines(int n, const float* P, const float* w0, const float* w1, int base_vertex, float* positions,
    float* normals, float* radius, int* lines, hipStream_t s) {
  long long threads = 5ll * n;
  hipLaunchKernelGGL(k_curves_to_lines, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, n, P, w0, w1, base_vertex,
      positions, normals, radius, lines);
  return (int)hipGetLastError();
}
int yhk_surface_lobe(int kind, int n, const float* params, const float* normal, const float* wo, const float* wi,
    const float* rn, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_surface_lobe, dim3((n + 255) / 256), dim3(256), 0, s, kind, n, params, normal, wo, wi, rn, out);
  return (int)hipGetLastError();
}
int yhk_surface_bsdf(int n, const void* mats, const float* normal, const float* wo, const float* wi, const float* rn,
    float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_surface_bsdf, dim3((n + 255) / 256), dim3(256), 0, s, n, (const yhd_material*)mats, normal, wo, wi, rn, out);
  return (int)hipGetLastError();
}
int yhk_hair_brdf(int n, const void* mats, const float* v, const float* nrm, const float* tng, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_hair_brdf, dim3((n + 255) / 256), dim3(256), 0, s, n, (const yh_material_in*)mats, v, nrm, tng, out);
  return (int)hipGetLastError();
}
int yhk_hair_eval(int n, const float* brdf, const float* wo, const float* wi, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_hair_eval, dim3((n + 63) / 64), dim3(256), 0, s, n, brdf, wo, wi, out);
  return (int)hipGetLastError();
}
int yhk_hair_pdf(int n, const float* brdf, const float* wo, const float* wi, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_hair_pdf, dim3((n + 63) / 64), dim3(256), 0, s, n, brdf, wo, wi, out);
  return (int)hipGetLastError();
}
int yhk_hair_sample(int n, const float* brdf, const float* wo, const float* rn, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_hair_sample, dim3((n + 255) / 256), dim3(256), 0, s, n, brdf, wo, rn, out);
  return (int)hipGetLastError();
}
int yhk_intersect(const yhd_scene* sc, int n, const float* rays, int* object, int* element, float* uv, float* dist,
    hipStream_t s) {
  hipLaunchKernelGGL(k_intersect, dim3((n + 63) / 64), dim3(256), 0, s, *sc, n, rays, object, element, uv, dist);
  return (int)hipGetLastError();
}
int yhk_selftest(int which, float beta_m, float beta_n, uint64_t state, uint64_t inc, int count, const float* wo,
    double* sums, unsigned int* worst_bits, hipStream_t s) {
  selftest_args a;
  a.which = which, a.beta_m = beta_m, a.beta_n = beta_n, a.state = state, a.inc = inc, a.count = count;
  a.wo[0] = wo[0], a.wo[1] = wo[1], a.wo[2] = wo[2];
  hipLaunchKernelGGL(k_selftest, dim3((count + 255) / 256), dim3(256), 0, s, a, sums, worst_bits);
  return (int)hipGetLastError();
}
}
