// yh_oracle.cpp — TEST INFRASTRUCTURE ONLY (see yh_oracle.h).
//
// Scalar CPU restatement of the hair path-tracing hot path of
// dsforza96/yocto-hair. Citations: ext.cpp = libs/yocto_extension/
// yocto_extension.cpp, pt.cpp = libs/yocto_pathtrace/yocto_pathtrace.cpp,
// math.h = libs/yocto/yocto_math.h. Float expression order follows the
// reference so that a g++ build is bit-identical to it (no -ffast-math, no
// FMA contraction on plain x86-64).
#include "yh_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// Small vector kit (math.h:1778-1812, 1979-2113): same operation order.
// ---------------------------------------------------------------------------
const float pif     = (float)3.14159265358979323846;  // math.h:214-215
const float flt_max = std::numeric_limits<float>::max();
const float flt_min = std::numeric_limits<float>::lowest();
const float flt_eps = std::numeric_limits<float>::epsilon();

inline float fabs_(float a) { return a < 0 ? -a : a; }          // math.h:1778
inline float fmin_(float a, float b) { return (a < b) ? a : b; }  // :1779
inline float fmax_(float a, float b) { return (a > b) ? a : b; }  // :1780
inline float fclamp(float a, float lo, float hi) {                // :1781
  return fmin_(fmax_(a, lo), hi);
}
inline int iclamp(int a, int lo, int hi) {
  return std::min(std::max(a, lo), hi);
}

struct V3 {
  float x, y, z;
};
inline V3    operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3    operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3    operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3    operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3    operator*(V3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline V3    operator*(float a, V3 b) { return {a * b.x, a * b.y, a * b.z}; }
inline V3    operator/(V3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline V3    operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3    operator+(V3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
inline V3    operator-(V3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
inline V3    operator-(float a, V3 b) { return {a - b.x, a - b.y, a - b.z}; }
inline bool  operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool  nonzero(V3 a) { return a.x || a.y || a.z; }  // math.h:1846
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3    cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3    normalize(V3 a) {  // math.h:2036-2039
  auto l = length(a);
  return (l != 0) ? a / l : a;
}
inline V3    orthonormalize(V3 a, V3 b) { return normalize(a - b * dot(a, b)); }
inline V3    vmin(V3 a, V3 b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
inline V3    vmax(V3 a, V3 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline float hmax(V3 a) { return fmax_(fmax_(a.x, a.y), a.z); }  // math.h:2092
inline float hmin(V3 a) { return fmin_(fmin_(a.x, a.y), a.z); }
inline bool  finite3(V3 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }
inline float luminance(V3 a) {  // math.h:3737-3739
  return (0.2126f * a.x + 0.7152f * a.y + 0.0722f * a.z);
}
inline V3 vexp(V3 a) { return {std::exp(a.x), std::exp(a.y), std::exp(a.z)}; }
inline V3 vlog(V3 a) { return {std::log(a.x), std::log(a.y), std::log(a.z)}; }
inline float at(V3 a, int i) { return (&a.x)[i]; }

struct Frame {
  V3 x, y, z, o;
};
struct BBox {
  V3 min = {flt_max, flt_max, flt_max};
  V3 max = {flt_min, flt_min, flt_min};
};
struct Ray {
  V3    o, d;
  float tmin = 1e-4f, tmax = flt_max;  // math.h:1106,1120-1124
};

inline Frame mkframe(const float* f) {
  return {{f[0], f[1], f[2]}, {f[3], f[4], f[5]}, {f[6], f[7], f[8]},
      {f[9], f[10], f[11]}};
}
// math.h:3136-3144
inline V3 transform_point(const Frame& a, V3 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.o;
}
inline V3 transform_vector(const Frame& a, V3 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
inline V3 transform_direction(const Frame& a, V3 b) {
  return normalize(transform_vector(a, b));
}
// math.h:2877-2885 with 2721-2741
inline Frame inverse(const Frame& a, bool non_rigid) {
  Frame r;
  if (non_rigid) {
    // adjoint = transpose({cross(y,z), cross(z,x), cross(x,y)}) * (1/det)
    auto c0  = cross(a.y, a.z), c1 = cross(a.z, a.x), c2 = cross(a.x, a.y);
    auto det = dot(a.x, cross(a.y, a.z));
    auto s   = 1 / det;
    r.x      = V3{c0.x, c1.x, c2.x} * s;
    r.y      = V3{c0.y, c1.y, c2.y} * s;
    r.z      = V3{c0.z, c1.z, c2.z} * s;
  } else {
    r.x = {a.x.x, a.y.x, a.z.x};
    r.y = {a.x.y, a.y.y, a.z.y};
    r.z = {a.x.z, a.y.z, a.z.z};
  }
  r.o = -(r.x * a.o.x + r.y * a.o.y + r.z * a.o.z);
  return r;
}
inline Ray transform_ray(const Frame& a, const Ray& b) {  // math.h:3160-3162
  return {transform_point(a, b.o), transform_vector(a, b.d), b.tmin, b.tmax};
}
inline BBox merge(const BBox& a, V3 b) { return {vmin(a.min, b), vmax(a.max, b)}; }
inline BBox merge(const BBox& a, const BBox& b) {
  return {vmin(a.min, b.min), vmax(a.max, b.max)};
}
inline V3   center(const BBox& a) { return (a.min + a.max) / 2; }
inline BBox transform_bbox(const Frame& a, const BBox& b) {  // math.h:3174-3185
  V3 corners[8] = {{b.min.x, b.min.y, b.min.z}, {b.min.x, b.min.y, b.max.z},
      {b.min.x, b.max.y, b.min.z}, {b.min.x, b.max.y, b.max.z},
      {b.max.x, b.min.y, b.min.z}, {b.max.x, b.min.y, b.max.z},
      {b.max.x, b.max.y, b.min.z}, {b.max.x, b.max.y, b.max.z}};
  auto x = BBox{};
  for (auto& c : corners) x = merge(x, transform_point(a, c));
  return x;
}

// ---------------------------------------------------------------------------
// PCG32 (math.h:1396-1442)
// ---------------------------------------------------------------------------
struct Rng {
  uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
};
inline uint32_t advance_rng(Rng& rng) {
  uint64_t old        = rng.state;
  rng.state           = old * 6364136223846793005ULL + rng.inc;
  uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot        = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
inline Rng make_rng(uint64_t seed, uint64_t seq = 1) {
  Rng rng;
  rng.state = 0U;
  rng.inc   = (seq << 1u) | 1u;
  advance_rng(rng);
  rng.state += seed;
  advance_rng(rng);
  return rng;
}
inline float rand1f(Rng& rng) {
  uint32_t u = (advance_rng(rng) >> 9) | 0x3f800000u;
  float    f;
  std::memcpy(&f, &u, 4);
  return f - 1.0f;
}
// rand1i(rng, 1 << 31): the int n is INT_MIN, converted to 2^31 for the
// unsigned modulo (math.h:1427, pt.cpp:1944)
inline int rand1i_2p31(Rng& rng) { return (int)(advance_rng(rng) % 2147483648u); }

// ---------------------------------------------------------------------------
// Sampling helpers (math.h:4847-4968)
// ---------------------------------------------------------------------------
inline V3 sample_sphere(float rx, float ry) {  // math.h:4847-4852
  auto z   = 2 * ry - 1;
  auto r   = std::sqrt(fclamp(1 - z * z, 0.0f, 1.0f));
  auto phi = 2 * pif * rx;
  return {r * std::cos(phi), r * std::sin(phi), z};
}
inline int sample_discrete_cdf(const std::vector<float>& cdf, float r) {
  // math.h:4957-4962
  r        = fclamp(r * cdf.back(), (float)0, cdf.back() - (float)0.00001);
  auto idx = (int)(std::upper_bound(cdf.data(), cdf.data() + cdf.size(), r) -
                   cdf.data());
  return iclamp(idx, 0, (int)cdf.size() - 1);
}
inline float sample_discrete_cdf_pdf(const std::vector<float>& cdf, int idx) {
  if (idx == 0) return cdf.at(0);  // math.h:4964-4967
  return cdf.at(idx) - cdf.at(idx - 1);
}

// ---------------------------------------------------------------------------
// Hair BSDF (ext.cpp:90-551)
// ---------------------------------------------------------------------------
const int   p_max           = 3;             // ext.h:84
const float sqrt_pi_over_8f = 0.626657069f;  // ext.cpp:90

struct HairMaterial {  // ext.h:86-95
  V3    sigma_a;
  float beta_m, beta_n, alpha, eta;
  V3    color;
  float eumelanin, pheomelanin;
};
struct HairBrdf {  // ext.h:97-113 (120 bytes)
  V3    sigma_a;
  float alpha, eta, h;
  float v[p_max + 1];
  float s;
  float sin_2k_alpha[3], cos_2k_alpha[3];
  float gamma_o;
  Frame world_to_brdf;
};
static_assert(sizeof(HairBrdf) == 120, "hair_brdf is 30 floats");
static_assert(sizeof(HairMaterial) == 48, "hair_material is 12 floats");

inline float sqr(float v) { return v * v; }
inline V3    sqr(V3 v) { return v * v; }
inline float powi(float v, int n) {  // ext.cpp:95-109 (template pow<n>)
  if (n == 0) return 1;
  if (n == 1) return v;
  auto n2 = powi(v, n / 2);
  return n2 * n2 * powi(v, n & 1);
}
// asin resolves to the global double overload in the reference (ext.cpp:111)
inline float safe_asin(float x) { return (float)::asin((double)fclamp(x, -1.0f, 1.0f)); }
inline float safe_sqrt(float x) { return std::sqrt(fmax_(0.0f, x)); }

HairBrdf eval_hair_brdf(const HairMaterial& m, float v, V3 normal, V3 tangent) {
  HairBrdf b{};
  b.sigma_a = {0, 0, 0};
  if (nonzero(m.sigma_a)) {  // ext.cpp:131-138
    b.sigma_a = m.sigma_a;
  } else if (nonzero(m.color)) {  // ext.cpp:121-125
    auto bn = m.beta_n;
    b.sigma_a = sqr(vlog(m.color) /
                    (5.969f - 0.215f * bn + 2.532f * sqr(bn) -
                        10.73f * powi(bn, 3) + 5.574f * powi(bn, 4) +
                        0.245f * powi(bn, 5)));
  } else if (m.eumelanin || m.pheomelanin) {  // ext.cpp:115-119
    b.sigma_a = m.eumelanin * V3{0.419f, 0.697f, 1.37f} +
                m.pheomelanin * V3{0.187f, 0.4f, 1.05f};
  }
  auto beta_m = m.beta_m, beta_n = m.beta_n;
  b.alpha   = m.alpha;
  b.eta     = m.eta;
  b.h       = -1 + 2 * v;  // ext.cpp:148
  b.gamma_o = safe_asin(b.h);
  b.v[0] = sqr(0.726f * beta_m + 0.812f * sqr(beta_m) + 3.7f * powi(beta_m, 20));
  b.v[1] = 0.25f * b.v[0];
  b.v[2] = 4 * b.v[0];
  b.v[3] = b.v[2];
  b.s    = sqrt_pi_over_8f *
        (0.265f * beta_n + 1.194f * sqr(beta_n) + 5.372f * powi(beta_n, 22));
  b.sin_2k_alpha[0] = std::sin(pif / 180 * b.alpha);
  b.cos_2k_alpha[0] = safe_sqrt(1 - sqr(b.sin_2k_alpha[0]));
  for (auto i = 1; i < 3; i++) {
    b.sin_2k_alpha[i] = 2 * b.cos_2k_alpha[i - 1] * b.sin_2k_alpha[i - 1];
    b.cos_2k_alpha[i] = sqr(b.cos_2k_alpha[i - 1]) - sqr(b.sin_2k_alpha[i - 1]);
  }
  // inverse(frame_fromzx(zero, normal, tangent)) (ext.cpp:174, math.h:2898)
  auto z = normalize(normal);
  auto x = orthonormalize(tangent, z);
  auto y = normalize(cross(z, x));
  b.world_to_brdf = inverse(Frame{x, y, z, {0, 0, 0}}, false);
  return b;
}

inline float i0(float x) {  // ext.cpp:179-192
  float   val   = 0;
  float   x2i   = 1;
  int64_t ifact = 1;
  int     i4    = 1;
  for (int i = 0; i < 10; i++) {
    if (i > 1) ifact *= i;
    val += x2i / (i4 * ifact * ifact);
    x2i *= x * x;
    i4 *= 4;
  }
  return val;
}
inline float log_i0(float x) {  // ext.cpp:194-199
  if (x > 12)
    return x + 0.5f * (-std::log(2 * pif) + std::log(1 / x) + 1 / (8 * x));
  else
    return std::log(i0(x));
}
float mp(float cos_theta_i, float cos_theta_o, float sin_theta_i,
    float sin_theta_o, float v) {  // ext.cpp:201-207
  auto a = cos_theta_i * cos_theta_o / v;
  auto b = sin_theta_i * sin_theta_o / v;
  // the second branch runs in double: sinh is the global double overload
  return (v <= 0.1f)
             ? (float)(double)(std::exp(log_i0(a) - b - 1 / v + 0.6931f +
                                        std::log(1 / (2 * v))))
             : (float)((double)(std::exp(-b) * i0(a)) /
                       (::sinh((double)(1 / v)) * 2 * v));
}
float fresnel_dielectric_cos(float eta, float cosw_) {  // math.h:4215-4235
  auto cosw  = fabs_(cosw_);
  auto sin2  = 1 - cosw * cosw;
  auto eta2  = eta * eta;
  auto cos2t = 1 - sin2 / eta2;
  if (cos2t < 0) return 1;
  auto t0 = std::sqrt(cos2t);
  auto t1 = eta * t0;
  auto t2 = eta * cosw;
  auto rs = (cosw - t1) / (cosw + t1);
  auto rp = (t0 - t2) / (t0 + t2);
  return (rs * rs + rp * rp) / 2;
}
void ap(float cos_theta_o, float eta, float h, V3 T, V3 out[p_max + 1]) {
  // ext.cpp:209-230
  auto cos_gamma_o = safe_sqrt(1 - h * h);
  auto cos_theta   = cos_theta_o * cos_gamma_o;
  // dot({0,0,1},{0,0,cos_theta}) = 0*0 + 0*0 + 1*cos_theta
  auto f = fresnel_dielectric_cos(eta, 0.0f * 0.0f + 0.0f * 0.0f + 1.0f * cos_theta);
  out[0] = {f, f, f};
  out[1] = sqr(1 - f) * T;
  for (auto p = 2; p < p_max; p++) out[p] = out[p - 1] * T * f;
  out[p_max] = out[p_max - 1] * f * T / (V3{1.f, 1.f, 1.f} - T * f);
}
inline float phi_fn(int p, float gamma_o, float gamma_t) {  // ext.cpp:232-234
  return 2 * p * gamma_t - 2 * gamma_o + p * pif;
}
inline float logistic(float x, float s) {  // ext.cpp:236-239
  x = fabs_(x);
  return std::exp(-x / s) / (s * sqr(1 + std::exp(-x / s)));
}
inline float logistic_cdf(float x, float s) { return 1 / (1 + std::exp(-x / s)); }
inline float trimmed_logistic(float x, float s, float a, float b) {
  return logistic(x, s) / (logistic_cdf(b, s) - logistic_cdf(a, s));
}
inline float np(float phi, int p, float s, float gamma_o, float gamma_t) {
  auto dphi = phi - phi_fn(p, gamma_o, gamma_t);  // ext.cpp:247-253
  while (dphi > pif) dphi -= 2 * pif;
  while (dphi < -pif) dphi += 2 * pif;
  return trimmed_logistic(dphi, s, -pif, pif);
}
inline void tilt(const HairBrdf& b, int p, float sin_theta_o, float cos_theta_o,
    float& sin_theta_op, float& cos_theta_op) {  // ext.cpp:299-322
  if (p == 0) {
    sin_theta_op = sin_theta_o * b.cos_2k_alpha[1] - cos_theta_o * b.sin_2k_alpha[1];
    cos_theta_op = cos_theta_o * b.cos_2k_alpha[1] + sin_theta_o * b.sin_2k_alpha[1];
  } else if (p == 1) {
    sin_theta_op = sin_theta_o * b.cos_2k_alpha[0] + cos_theta_o * b.sin_2k_alpha[0];
    cos_theta_op = cos_theta_o * b.cos_2k_alpha[0] - sin_theta_o * b.sin_2k_alpha[0];
  } else if (p == 2) {
    sin_theta_op = sin_theta_o * b.cos_2k_alpha[2] + cos_theta_o * b.sin_2k_alpha[2];
    cos_theta_op = cos_theta_o * b.cos_2k_alpha[2] - sin_theta_o * b.sin_2k_alpha[2];
  } else {
    sin_theta_op = sin_theta_o;
    cos_theta_op = cos_theta_o;
  }
}

V3 eval_hair_scattering(const HairBrdf& b, V3 outgoing_, V3 incoming_) {
  // ext.cpp:255-336
  auto outgoing    = transform_direction(b.world_to_brdf, outgoing_);
  auto incoming    = transform_direction(b.world_to_brdf, incoming_);
  auto sin_theta_o = outgoing.x;
  auto cos_theta_o = safe_sqrt(1 - sqr(sin_theta_o));
  auto phi_o       = std::atan2(outgoing.z, outgoing.y);
  auto sin_theta_i = incoming.x;
  auto cos_theta_i = safe_sqrt(1 - sqr(sin_theta_i));
  auto phi_i       = std::atan2(incoming.z, incoming.y);
  auto sin_theta_t = sin_theta_o / b.eta;
  auto cos_theta_t = safe_sqrt(1 - sqr(sin_theta_t));
  auto etap        = std::sqrt(b.eta * b.eta - sqr(sin_theta_o)) / cos_theta_o;
  auto sin_gamma_t = b.h / etap;
  auto cos_gamma_t = safe_sqrt(1 - sqr(sin_gamma_t));
  auto gamma_t     = safe_asin(sin_gamma_t);
  auto T           = vexp(-b.sigma_a * (2 * cos_gamma_t / cos_theta_t));
  auto phi         = phi_i - phi_o;
  V3   apv[p_max + 1];
  ap(cos_theta_o, b.eta, b.h, T, apv);
  auto fsum = V3{0, 0, 0};
  for (auto p = 0; p < p_max; p++) {
    float sin_theta_op, cos_theta_op;
    tilt(b, p, sin_theta_o, cos_theta_o, sin_theta_op, cos_theta_op);
    cos_theta_op = fabs_(cos_theta_op);
    fsum = fsum + mp(cos_theta_i, cos_theta_op, sin_theta_i, sin_theta_op, b.v[p]) *
                      apv[p] * np(phi, p, b.s, b.gamma_o, gamma_t);
  }
  fsum = fsum + mp(cos_theta_i, cos_theta_o, sin_theta_i, sin_theta_o, b.v[p_max]) *
                    apv[p_max] / (2 * pif);
  return fsum;
}

inline uint32_t compact1by1(uint32_t x) {  // ext.cpp:339-351
  x &= 0x55555555;
  x = (x ^ (x >> 1)) & 0x33333333;
  x = (x ^ (x >> 2)) & 0x0f0f0f0f;
  x = (x ^ (x >> 4)) & 0x00ff00ff;
  x = (x ^ (x >> 8)) & 0x0000ffff;
  return x;
}
inline void demux_float(float f, float out[2]) {  // ext.cpp:353-357
  uint64_t v       = f * (1ull << 32);
  uint32_t bits[2] = {compact1by1(v), compact1by1(v >> 1)};
  out[0] = bits[0] / float(1 << 16), out[1] = bits[1] / float(1 << 16);
}
inline float sample_trimmed_logistic(float u, float s, float a, float b) {
  auto k = logistic_cdf(b, s) - logistic_cdf(a, s);  // ext.cpp:359-363
  auto x = -s * std::log(1 / (u * k + logistic_cdf(a, s)) - 1);
  return fclamp(x, a, b);
}
void compute_ap_pdf(const HairBrdf& b, float cos_theta_o, float ap_pdf[p_max + 1]) {
  // ext.cpp:365-397
  auto sin_theta_o = safe_sqrt(1 - cos_theta_o * cos_theta_o);
  auto sin_theta_t = sin_theta_o / b.eta;
  auto cos_theta_t = safe_sqrt(1 - sqr(sin_theta_t));
  auto etap        = std::sqrt(b.eta * b.eta - sqr(sin_theta_o)) / cos_theta_o;
  auto sin_gamma_t = b.h / etap;
  auto cos_gamma_t = safe_sqrt(1 - sqr(sin_gamma_t));
  auto T           = vexp(-b.sigma_a * (2 * cos_gamma_t / cos_theta_t));
  V3   apv[p_max + 1];
  ap(cos_theta_o, b.eta, b.h, T, apv);
  auto sum_y = 0.0f;
  for (auto i = 0; i <= p_max; i++) sum_y += luminance(apv[i]);
  for (auto i = 0; i <= p_max; i++) ap_pdf[i] = luminance(apv[i]) / sum_y;
}

V3 sample_hair_scattering(const HairBrdf& b, V3 outgoing_, float rnx, float rny) {
  // ext.cpp:399-479
  auto  outgoing    = transform_direction(b.world_to_brdf, outgoing_);
  auto  sin_theta_o = outgoing.x;
  auto  cos_theta_o = safe_sqrt(1 - sqr(sin_theta_o));
  auto  phi_o       = std::atan2(outgoing.z, outgoing.y);
  float u[2][2];
  demux_float(rnx, u[0]);
  demux_float(rny, u[1]);
  float ap_pdf[p_max + 1];
  compute_ap_pdf(b, cos_theta_o, ap_pdf);
  auto p = 0;
  for (p = 0; p < p_max; p++) {
    if (u[0][0] < ap_pdf[p]) break;
    u[0][0] -= ap_pdf[p];
  }
  float sin_theta_op, cos_theta_op;
  tilt(b, p, sin_theta_o, cos_theta_o, sin_theta_op, cos_theta_op);
  u[1][0]        = fmax_(u[1][0], 1e-5f);
  auto cos_theta = 1 + b.v[p] * std::log(u[1][0] + (1 - u[1][0]) * std::exp(-2 / b.v[p]));
  auto sin_theta = safe_sqrt(1 - sqr(cos_theta));
  auto cos_phi   = std::cos(2 * pif * u[1][1]);
  auto sin_theta_i = -cos_theta * sin_theta_op + sin_theta * cos_phi * cos_theta_op;
  auto cos_theta_i = safe_sqrt(1 - sqr(sin_theta_i));
  auto etap        = std::sqrt(b.eta * b.eta - sqr(sin_theta_o)) / cos_theta_o;
  auto sin_gamma_t = b.h / etap;
  auto gamma_t     = safe_asin(sin_gamma_t);
  auto dphi        = 0.0f;
  if (p < p_max)
    dphi = phi_fn(p, b.gamma_o, gamma_t) +
           sample_trimmed_logistic(u[0][1], b.s, -pif, pif);
  else
    dphi = 2 * pif * u[0][1];
  auto phi_i    = phi_o + dphi;
  auto incoming = V3{sin_theta_i, cos_theta_i * std::cos(phi_i),
      cos_theta_i * std::sin(phi_i)};
  return transform_direction(inverse(b.world_to_brdf, false), incoming);
}

float sample_hair_scattering_pdf(const HairBrdf& b, V3 outgoing_, V3 incoming_) {
  // ext.cpp:481-551
  auto outgoing    = transform_direction(b.world_to_brdf, outgoing_);
  auto incoming    = transform_direction(b.world_to_brdf, incoming_);
  auto sin_theta_o = outgoing.x;
  auto cos_theta_o = safe_sqrt(1 - sqr(sin_theta_o));
  auto phi_o       = std::atan2(outgoing.z, outgoing.y);
  auto sin_theta_i = incoming.x;
  auto cos_theta_i = safe_sqrt(1 - sqr(sin_theta_i));
  auto phi_i       = std::atan2(incoming.z, incoming.y);
  auto etap        = std::sqrt(b.eta * b.eta - sqr(sin_theta_o)) / cos_theta_o;
  auto sin_gamma_t = b.h / etap;
  auto gamma_t     = safe_asin(sin_gamma_t);
  float ap_pdf[p_max + 1];
  compute_ap_pdf(b, cos_theta_o, ap_pdf);
  auto phi = phi_i - phi_o;
  auto pdf = 0.0f;
  for (auto p = 0; p < p_max; p++) {
    float sin_theta_op, cos_theta_op;
    tilt(b, p, sin_theta_o, cos_theta_o, sin_theta_op, cos_theta_op);
    cos_theta_op = fabs_(cos_theta_op);
    pdf += mp(cos_theta_i, cos_theta_op, sin_theta_i, sin_theta_op, b.v[p]) *
           ap_pdf[p] * np(phi, p, b.s, b.gamma_o, gamma_t);
  }
  pdf += mp(cos_theta_i, cos_theta_o, sin_theta_i, sin_theta_o, b.v[p_max]) *
         ap_pdf[p_max] * (1 / (2 * pif));
  return pdf;
}

// ---------------------------------------------------------------------------
// Intersection primitives (math.h:3426-3505, 3544-3554)
// ---------------------------------------------------------------------------
inline bool intersect_line(const Ray& ray, V3 p0, V3 p1, float r0, float r1,
    float uv[2], float& dist) {
  auto u = ray.d, v = p1 - p0, w = ray.o - p0;
  auto a = dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w);
  auto det = a * c - b * b;
  if (det == 0) return false;
  auto t = (b * e - c * d) / det;
  auto s = (a * e - b * d) / det;
  if (t < ray.tmin || t > ray.tmax) return false;
  s        = fclamp(s, (float)0, (float)1);
  auto pr  = ray.o + ray.d * t;
  auto pl  = p0 + (p1 - p0) * s;
  auto prl = pr - pl;
  auto d2  = dot(prl, prl);
  auto r   = r0 * (1 - s) + r1 * s;
  if (d2 > r * r) return false;
  uv[0] = s, uv[1] = std::sqrt(d2) / r;
  dist = t;
  return true;
}
inline bool intersect_triangle(const Ray& ray, V3 p0, V3 p1, V3 p2, float uv[2],
    float& dist) {
  auto edge1 = p1 - p0, edge2 = p2 - p0;
  auto pvec = cross(ray.d, edge2);
  auto det  = dot(edge1, pvec);
  if (det == 0) return false;
  auto inv_det = 1.0f / det;
  auto tvec    = ray.o - p0;
  auto u       = dot(tvec, pvec) * inv_det;
  if (u < 0 || u > 1) return false;
  auto qvec = cross(tvec, edge1);
  auto v    = dot(ray.d, qvec) * inv_det;
  if (v < 0 || u + v > 1) return false;
  auto t = dot(edge2, qvec) * inv_det;
  if (t < ray.tmin || t > ray.tmax) return false;
  uv[0] = u, uv[1] = v;
  dist = t;
  return true;
}
inline bool intersect_bbox(const Ray& ray, V3 ray_dinv, const BBox& bbox) {
  auto it_min = (bbox.min - ray.o) * ray_dinv;
  auto it_max = (bbox.max - ray.o) * ray_dinv;
  auto tmin   = vmin(it_min, it_max);
  auto tmax   = vmax(it_min, it_max);
  auto t0     = fmax_(hmax(tmin), ray.tmin);
  auto t1     = fmin_(hmin(tmax), ray.tmax);
  t1 *= 1.00000024f;
  return t0 <= t1;
}

// ---------------------------------------------------------------------------
// Scene, BVH build (pt.cpp:557-818) and traversal (pt.cpp:821-1053)
// ---------------------------------------------------------------------------
struct BvhNode {  // pt.h:243-249 (32 bytes)
  BBox          bbox;
  int           start;
  short         num;
  bool          internal;
  unsigned char axis;
};
static_assert(sizeof(BvhNode) == 32, "bvh_node layout");
struct BvhTree {
  std::vector<BvhNode> nodes;
  std::vector<int>     primitives;
};
struct BvhPrim {
  BBox bbox;
  V3   center;
  int  primitive;
};

struct Shape {
  std::vector<V3>    positions, normals;
  std::vector<float> radius, texcoords;  // texcoords: 2 per vertex or empty
  std::vector<int>   lines, triangles;  // flattened pairs / triples
  BvhTree            bvh;
  int nlines() const { return (int)lines.size() / 2; }
  int ntriangles() const { return (int)triangles.size() / 3; }
};
struct Material {
  V3           emission, color;
  float        specular = 0, metallic = 0, roughness = 0, ior = 1.5f, transmission = 0, opacity = 1;
  V3           scattering{0, 0, 0};
  float        scanisotropy = 0, trdepth = 0.01f;
  int          emission_tex = -1, color_tex = -1, scattering_tex = -1;  // index into yo_scene::textures
  HairMaterial hair;
  bool         thin;
};
struct Texture {  // ptr::texture colorf / colorb (pt.h:282-287)
  int                        w = 0, h = 0;
  std::vector<V3>            colorf;
  std::vector<unsigned char> colorb;  // RGB
};
struct Object {
  Frame frame;
  int   shape, material;
};
struct Environment {
  Frame           frame;
  V3              emission;
  int             w = 0, h = 0;
  std::vector<V3> texels;
};
struct Light {
  int                object = -1, environment = -1;
  std::vector<float> cdf;
};
struct Camera {
  Frame frame;
  float lens, film[2], focus, aperture;
};

struct Counters {
  uint64_t rays = 0, nodes = 0, seg = 0, tri = 0, hair = 0, surf = 0, envl = 0,
           envs = 0, samples = 0;
};
thread_local Counters tls_counters;

std::pair<int, int> split_middle(std::vector<BvhPrim>& prims, int start, int end) {
  // pt.cpp:564-595
  auto axis  = 0;
  auto mid   = (start + end) / 2;
  auto cbbox = BBox{};
  for (auto i = start; i < end; i++) cbbox = merge(cbbox, prims[i].center);
  auto csize = cbbox.max - cbbox.min;
  if (csize == V3{0, 0, 0}) return {mid, axis};
  if (csize.x >= csize.y && csize.x >= csize.z) axis = 0;
  if (csize.y >= csize.x && csize.y >= csize.z) axis = 1;
  if (csize.z >= csize.x && csize.z >= csize.y) axis = 2;
  auto middle = at(center(cbbox), axis);
  mid = (int)(std::partition(prims.data() + start, prims.data() + end,
                  [axis, middle](const BvhPrim& p) { return at(p.center, axis) < middle; }) -
              prims.data());
  if (mid == start || mid == end) mid = (start + end) / 2;
  return {mid, axis};
}
const int bvh_max_prims = 4;  // pt.cpp:598
void build_bvh(BvhTree& tree, std::vector<BvhPrim>& prims) {  // pt.cpp:601-650
  auto& nodes = tree.nodes;
  nodes.clear();
  nodes.reserve(prims.size() * 2);
  struct Item { int node, start, end; };
  auto queue = std::deque<Item>{{0, 0, (int)prims.size()}};
  nodes.emplace_back();
  while (!queue.empty()) {
    auto next = queue.front();
    queue.pop_front();
    auto nodeid = next.node, start = next.start, end = next.end;
    auto& node = nodes[nodeid];
    node.bbox  = BBox{};
    for (auto i = start; i < end; i++) node.bbox = merge(node.bbox, prims[i].bbox);
    if (end - start > bvh_max_prims) {
      auto [mid, axis] = split_middle(prims, start, end);
      node.internal = true;
      node.axis     = (unsigned char)axis;
      node.num      = 2;
      node.start    = (int)nodes.size();
      auto first    = node.start;  // (node reference dies on emplace_back)
      nodes.emplace_back();
      nodes.emplace_back();
      queue.push_back({first + 0, start, mid});
      queue.push_back({first + 1, mid, end});
    } else {
      node.internal = false;
      node.num      = (short)(end - start);
      node.start    = start;
      node.axis     = 0;
    }
  }
  nodes.shrink_to_fit();
  tree.primitives.clear();
  tree.primitives.reserve(prims.size());
  for (auto& p : prims) tree.primitives.push_back(p.primitive);
}
void init_shape_bvh(Shape& shape) {  // pt.cpp:713-752
  auto prims = std::vector<BvhPrim>{};
  if (shape.nlines()) {
    for (auto idx = 0; idx < shape.nlines(); idx++) {
      auto l0 = shape.lines[2 * idx], l1 = shape.lines[2 * idx + 1];
      auto p0 = shape.positions[l0], p1 = shape.positions[l1];
      auto r0 = shape.radius[l0], r1 = shape.radius[l1];
      BvhPrim p;
      p.bbox      = {vmin(p0 - r0, p1 - r1), vmax(p0 + r0, p1 + r1)};  // math.h:3037
      p.center    = center(p.bbox);
      p.primitive = idx;
      prims.push_back(p);
    }
  } else if (shape.ntriangles()) {
    for (auto idx = 0; idx < shape.ntriangles(); idx++) {
      auto p0 = shape.positions[shape.triangles[3 * idx]];
      auto p1 = shape.positions[shape.triangles[3 * idx + 1]];
      auto p2 = shape.positions[shape.triangles[3 * idx + 2]];
      BvhPrim p;
      p.bbox      = {vmin(p0, vmin(p1, p2)), vmax(p0, vmax(p1, p2))};  // :3041
      p.center    = center(p.bbox);
      p.primitive = idx;
      prims.push_back(p);
    }
  }
  build_bvh(shape.bvh, prims);
}

}  // namespace

struct yo_scene {
  std::vector<Texture>     textures;
  std::vector<Shape>       shapes;
  std::vector<Material>    materials;
  std::vector<Object>      objects;
  std::vector<Environment> environments;
  std::vector<Light>       lights;
  Camera                   camera;
  BvhTree                  bvh;
};

namespace {

bool intersect_shape_bvh(const Shape& shape, const Ray& ray_, int& element,
    float uv[2], float& distance) {  // pt.cpp:821-931 (find_any = false)
  auto& bvh = shape.bvh;
  if (bvh.nodes.empty()) return false;
  int  node_stack[128];
  auto node_cur          = 0;
  node_stack[node_cur++] = 0;
  auto hit               = false;
  auto ray               = ray_;
  auto ray_dinv  = V3{1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z};
  int  ray_dsign[3] = {(ray_dinv.x < 0) ? 1 : 0, (ray_dinv.y < 0) ? 1 : 0,
      (ray_dinv.z < 0) ? 1 : 0};
  auto& cnt = tls_counters;
  while (node_cur) {
    auto& node = bvh.nodes[node_stack[--node_cur]];
    cnt.nodes++;
    if (!intersect_bbox(ray, ray_dinv, node.bbox)) continue;
    if (node.internal) {
      if (ray_dsign[node.axis]) {
        node_stack[node_cur++] = node.start + 0;
        node_stack[node_cur++] = node.start + 1;
      } else {
        node_stack[node_cur++] = node.start + 1;
        node_stack[node_cur++] = node.start + 0;
      }
    } else if (shape.nlines()) {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto prim = bvh.primitives[idx];
        auto l0 = shape.lines[2 * prim], l1 = shape.lines[2 * prim + 1];
        cnt.seg++;
        if (intersect_line(ray, shape.positions[l0], shape.positions[l1],
                shape.radius[l0], shape.radius[l1], uv, distance)) {
          hit      = true;
          element  = prim;
          ray.tmax = distance;
        }
      }
    } else if (shape.ntriangles()) {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto prim = bvh.primitives[idx];
        cnt.tri++;
        if (intersect_triangle(ray, shape.positions[shape.triangles[3 * prim]],
                shape.positions[shape.triangles[3 * prim + 1]],
                shape.positions[shape.triangles[3 * prim + 2]], uv, distance)) {
          hit      = true;
          element  = prim;
          ray.tmax = distance;
        }
      }
    }
  }
  return hit;
}

bool intersect_instance_bvh(const yo_scene& scene, int object, const Ray& ray,
    int& element, float uv[2], float& distance) {  // pt.cpp:1031-1037
  auto& obj     = scene.objects[object];
  auto  inv_ray = transform_ray(inverse(obj.frame, true), ray);
  return intersect_shape_bvh(scene.shapes[obj.shape], inv_ray, element, uv, distance);
}

bool intersect_scene_bvh(const yo_scene& scene, const Ray& ray_, int& object,
    int& element, float uv[2], float& distance) {  // pt.cpp:934-1028
  auto& bvh = scene.bvh;
  if (bvh.nodes.empty()) return false;
  int  node_stack[128];
  auto node_cur          = 0;
  node_stack[node_cur++] = 0;
  auto hit               = false;
  auto ray               = ray_;
  auto ray_dinv  = V3{1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z};
  int  ray_dsign[3] = {(ray_dinv.x < 0) ? 1 : 0, (ray_dinv.y < 0) ? 1 : 0,
      (ray_dinv.z < 0) ? 1 : 0};
  tls_counters.rays++;
  while (node_cur) {
    auto& node = bvh.nodes[node_stack[--node_cur]];
    tls_counters.nodes++;
    if (!intersect_bbox(ray, ray_dinv, node.bbox)) continue;
    if (node.internal) {
      if (ray_dsign[node.axis]) {
        node_stack[node_cur++] = node.start + 0;
        node_stack[node_cur++] = node.start + 1;
      } else {
        node_stack[node_cur++] = node.start + 1;
        node_stack[node_cur++] = node.start + 0;
      }
    } else {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto oid = bvh.primitives[idx];
        if (intersect_instance_bvh(scene, oid, ray, element, uv, distance)) {
          hit      = true;
          object   = oid;
          ray.tmax = distance;
        }
      }
    }
  }
  return hit;
}

// ---------------------------------------------------------------------------
// Shading-point evaluation (pt.cpp:232-492)
// ---------------------------------------------------------------------------
V3 eval_position(const yo_scene& scene, int object, int element, const float uv[2]) {
  auto& obj   = scene.objects[object];  // pt.cpp:232-250
  auto& shape = scene.shapes[obj.shape];
  if (shape.ntriangles()) {
    auto p0 = shape.positions[shape.triangles[3 * element]];
    auto p1 = shape.positions[shape.triangles[3 * element + 1]];
    auto p2 = shape.positions[shape.triangles[3 * element + 2]];
    return transform_point(obj.frame, p0 * (1 - uv[0] - uv[1]) + p1 * uv[0] + p2 * uv[1]);
  } else if (shape.nlines()) {
    auto p0 = shape.positions[shape.lines[2 * element]];
    auto p1 = shape.positions[shape.lines[2 * element + 1]];
    return transform_point(obj.frame, p0 * (1 - uv[0]) + p1 * uv[0]);
  }
  return {0, 0, 0};
}
// eval_texcoord (pt.cpp:295-311): interpolate_triangle / interpolate_line (math.h:3322-3331)
void eval_texcoord(const yo_scene& scene, int object, int element, const float uv[2], float tc[2]) {
  auto& obj   = scene.objects[object];
  auto& shape = scene.shapes[obj.shape];
  if (shape.texcoords.empty()) {
    tc[0] = uv[0], tc[1] = uv[1];
  } else if (shape.ntriangles()) {
    auto t = &shape.triangles[3 * element];
    for (int k = 0; k < 2; k++)
      tc[k] = shape.texcoords[2 * t[0] + k] * (1 - uv[0] - uv[1]) + shape.texcoords[2 * t[1] + k] * uv[0] +
              shape.texcoords[2 * t[2] + k] * uv[1];
  } else if (shape.nlines()) {
    auto l = &shape.lines[2 * element];
    for (int k = 0; k < 2; k++) tc[k] = shape.texcoords[2 * l[0] + k] * (1 - uv[0]) + shape.texcoords[2 * l[1] + k] * uv[0];
  } else {
    tc[0] = tc[1] = 0;
  }
}
inline float srgb_to_rgb(float srgb) {  // math.h:3742-3745 (the comparison is in double)
  return (srgb <= 0.04045) ? srgb / 12.92f : std::pow((srgb + 0.055f) / (1.0f + 0.055f), 2.4f);
}
V3 lookup_texture(const Texture& t, int i, int j, bool ldr_as_linear) {  // pt.cpp:147-164
  if (!t.colorf.empty()) return t.colorf[(size_t)j * t.w + i];
  auto b = &t.colorb[((size_t)j * t.w + i) * 3];
  auto f = V3{b[0] / 255.0f, b[1] / 255.0f, b[2] / 255.0f};  // byte_to_float (math.h:3718-3730)
  return ldr_as_linear ? f : V3{srgb_to_rgb(f.x), srgb_to_rgb(f.y), srgb_to_rgb(f.z)};
}
V3 eval_texture(const Texture* t, const float uv[2], bool ldr_as_linear = false) {  // pt.cpp:167-200
  if (!t) return {1, 1, 1};
  auto sx = t->w, sy = t->h;
  auto s = std::fmod(uv[0], 1.0f) * sx;
  if (s < 0) s += sx;
  auto tt = std::fmod(uv[1], 1.0f) * sy;
  if (tt < 0) tt += sy;
  auto i = iclamp((int)s, 0, sx - 1), j = iclamp((int)tt, 0, sy - 1);
  auto ii = (i + 1) % sx, jj = (j + 1) % sy;
  auto u = s - i, v = tt - j;
  return lookup_texture(*t, i, j, ldr_as_linear) * (1 - u) * (1 - v) + lookup_texture(*t, i, jj, ldr_as_linear) * (1 - u) * v +
         lookup_texture(*t, ii, j, ldr_as_linear) * u * (1 - v) + lookup_texture(*t, ii, jj, ldr_as_linear) * u * v;
}
inline const Texture* texture_of(const yo_scene& scene, int id) { return id >= 0 ? &scene.textures[(size_t)id] : nullptr; }

// transform_normal(frame, n) with non_rigid = false (math.h:3145-3152)
V3 transform_normal(const Frame& a, V3 b) { return normalize(transform_vector(a, b)); }
V3 eval_element_normal(const yo_scene& scene, int object, int element) {
  auto& obj   = scene.objects[object];  // pt.cpp:253-269
  auto& shape = scene.shapes[obj.shape];
  if (shape.ntriangles()) {
    auto p0 = shape.positions[shape.triangles[3 * element]];
    auto p1 = shape.positions[shape.triangles[3 * element + 1]];
    auto p2 = shape.positions[shape.triangles[3 * element + 2]];
    return transform_normal(obj.frame, normalize(cross(p1 - p0, p2 - p0)));
  } else if (shape.nlines()) {
    auto p0 = shape.positions[shape.lines[2 * element]];
    auto p1 = shape.positions[shape.lines[2 * element + 1]];
    return transform_normal(obj.frame, normalize(p1 - p0));
  }
  return {0, 0, 0};
}
V3 eval_normal(const yo_scene& scene, int object, int element, const float uv[2]) {
  auto& obj   = scene.objects[object];  // pt.cpp:272-292
  auto& shape = scene.shapes[obj.shape];
  if (shape.normals.empty()) return eval_element_normal(scene, object, element);
  if (shape.ntriangles()) {
    auto n0 = shape.normals[shape.triangles[3 * element]];
    auto n1 = shape.normals[shape.triangles[3 * element + 1]];
    auto n2 = shape.normals[shape.triangles[3 * element + 2]];
    return transform_normal(
        obj.frame, normalize(n0 * (1 - uv[0] - uv[1]) + n1 * uv[0] + n2 * uv[1]));
  } else if (shape.nlines()) {
    auto n0 = shape.normals[shape.lines[2 * element]];
    auto n1 = shape.normals[shape.lines[2 * element + 1]];
    return transform_normal(obj.frame, normalize(n0 * (1 - uv[0]) + n1 * uv[0]));
  }
  return {0, 0, 0};
}
V3 eval_shading_normal(const yo_scene& scene, int object, int element,
    const float uv[2], V3 outgoing) {  // pt.cpp:350-369
  auto& obj   = scene.objects[object];
  auto& shape = scene.shapes[obj.shape];
  if (shape.ntriangles()) {
    auto normal = eval_normal(scene, object, element, uv);
    if (!scene.materials[obj.material].thin) return normal;
    return dot(normal, outgoing) >= 0 ? normal : -normal;
  } else if (shape.nlines()) {
    auto normal = eval_normal(scene, object, element, uv);
    return orthonormalize(outgoing, normal);
  }
  return {0, 0, 0};
}

// ---------------------------------------------------------------------------
// Surface lobes (math.h:4215-4755, 2059-2067). Only the GGX forms the
// reference's defaults select are restated. Operation order is the
// reference's: with F held as a vector the dielectric ({1,1,1} * F * ...) and
// conductor (F * ...) expressions are the same per-channel arithmetic.
// ---------------------------------------------------------------------------
inline V3 reflect(V3 w, V3 n) { return -w + 2 * dot(n, w) * n; }  // math.h:2059
inline V3 refract(V3 w, V3 n, float inv_eta) {                    // math.h:2062
  auto cosine = dot(n, w);
  auto k      = 1 + inv_eta * inv_eta * (cosine * cosine - 1);
  if (k < 0) return {0, 0, 0};
  return -w * inv_eta + (inv_eta * cosine - std::sqrt(k)) * n;
}
inline float fresnel_dielectric(float eta, V3 normal, V3 outgoing) {  // math.h:4212
  return fresnel_dielectric_cos(eta, dot(normal, outgoing));
}
float conductor_channel(float eta, float etak, float cosw, float cos2, float sin2) {
  auto eta2 = eta * eta, etak2 = etak * etak;  // math.h:4246-4260, one channel
  auto t0       = eta2 - etak2 - sin2;
  auto a2plusb2 = std::sqrt(t0 * t0 + 4 * eta2 * etak2);
  auto t1       = a2plusb2 + cos2;
  auto a        = std::sqrt((a2plusb2 + t0) / 2);
  auto t2       = 2 * a * cosw;
  auto rs       = (t1 - t2) / (t1 + t2);
  auto t3       = cos2 * a2plusb2 + sin2 * sin2;
  auto t4       = t2 * sin2;
  auto rp       = rs * (t3 - t4) / (t3 + t4);
  return (rp + rs) / 2;
}
V3 fresnel_conductor(V3 eta, V3 etak, V3 normal, V3 outgoing) {  // math.h:4238-4262
  auto cosw = dot(normal, outgoing);
  if (cosw <= 0) return {0, 0, 0};
  cosw      = fclamp(cosw, (float)-1, (float)1);
  auto cos2 = cosw * cosw;
  auto sin2 = fclamp(1 - cos2, (float)0, (float)1);
  return {conductor_channel(eta.x, etak.x, cosw, cos2, sin2),
      conductor_channel(eta.y, etak.y, cosw, cos2, sin2),
      conductor_channel(eta.z, etak.z, cosw, cos2, sin2)};
}
V3 reflectivity_to_eta(V3 reflectivity) {  // math.h:4270-4273
  auto one = [](float r_) {
    auto r = fclamp(r_, 0.0f, 0.99f);
    return (1 + std::sqrt(r)) / (1 - std::sqrt(r));
  };
  return {one(reflectivity.x), one(reflectivity.y), one(reflectivity.z)};
}
float microfacet_distribution(float roughness, V3 normal, V3 halfway) {  // math.h:4307-4322
  auto cosine = dot(normal, halfway);
  if (cosine <= 0) return 0;
  auto roughness2 = roughness * roughness;
  auto cosine2    = cosine * cosine;
  return roughness2 / (pif * (cosine2 * roughness2 + 1 - cosine2) *
                          (cosine2 * roughness2 + 1 - cosine2));
}
float microfacet_shadowing1(float roughness, V3 normal, V3 halfway, V3 direction) {
  auto cosine  = dot(normal, direction);  // math.h:4325-4343
  auto cosineh = dot(halfway, direction);
  if (cosine * cosineh <= 0) return 0;
  auto roughness2 = roughness * roughness;
  auto cosine2    = cosine * cosine;
  return 2 * fabs_(cosine) /
         (fabs_(cosine) + std::sqrt(cosine2 - roughness2 * cosine2 + roughness2));
}
float microfacet_shadowing(float roughness, V3 normal, V3 halfway, V3 outgoing, V3 incoming) {
  return microfacet_shadowing1(roughness, normal, halfway, outgoing) *  // math.h:4346-4351
         microfacet_shadowing1(roughness, normal, halfway, incoming);
}
// basis_fromz + transform_direction(mat3f, v) (math.h:2743-2752)
V3 from_local_z(V3 normal, V3 local) {
  auto zz   = normalize(normal);
  auto sign = copysignf(1.0f, zz.z);
  auto a    = -1.0f / (sign + zz.z);
  auto b    = zz.x * zz.y * a;
  auto x    = V3{1.0f + sign * zz.x * zz.x * a, sign * b, -sign * zz.x};
  auto y    = V3{b, sign + zz.y * zz.y * a, -zz.y};
  return normalize(x * local.x + y * local.y + zz * local.z);
}
V3 sample_microfacet(float roughness, V3 normal, float rx, float ry) {  // math.h:4354-4367
  auto phi   = 2 * pif * rx;
  auto theta = std::atan(roughness * std::sqrt(ry / (1 - ry)));
  return from_local_z(normal, {std::cos(phi) * std::sin(theta),
                                  std::sin(phi) * std::sin(theta), std::cos(theta)});
}
float sample_microfacet_pdf(float roughness, V3 normal, V3 halfway) {  // math.h:4370-4375
  auto cosine = dot(normal, halfway);
  if (cosine < 0) return 0;
  return microfacet_distribution(roughness, normal, halfway) * cosine;
}
V3 sample_hemisphere_cos(V3 normal, float rx, float ry) {  // math.h:4856-4878
  auto z   = std::sqrt(ry);
  auto r   = std::sqrt(1 - z * z);
  auto phi = 2 * pif * rx;
  return from_local_z(normal, {r * std::cos(phi), r * std::sin(phi), z});
}
inline bool same_side_up(V3 normal, V3 outgoing, V3 incoming) {
  return !(dot(normal, incoming) <= 0 || dot(normal, outgoing) <= 0);
}

// --- rough lobes: value * |cos| --------------------------------------------
V3 eval_diffuse_reflection(V3 normal, V3 outgoing, V3 incoming) {  // math.h:4427-4431
  if (!same_side_up(normal, outgoing, incoming)) return {0, 0, 0};
  return V3{1, 1, 1} / pif * dot(normal, incoming);
}
// math.h:4441-4466 with F = {f,f,f} (dielectric) or the conductor Fresnel
V3 microfacet_reflection_with(V3 F, float roughness, V3 normal, V3 halfway, V3 outgoing,
    V3 incoming) {
  auto D = microfacet_distribution(roughness, normal, halfway);
  auto G = microfacet_shadowing(roughness, normal, halfway, outgoing, incoming);
  return F * D * G / (4 * dot(normal, outgoing) * dot(normal, incoming)) *
         dot(normal, incoming);
}
V3 eval_microfacet_reflection(float ior, float roughness, V3 normal, V3 outgoing, V3 incoming) {
  if (!same_side_up(normal, outgoing, incoming)) return {0, 0, 0};
  auto halfway = normalize(incoming + outgoing);
  auto f       = fresnel_dielectric(ior, halfway, incoming);
  return microfacet_reflection_with(V3{1, 1, 1} * f, roughness, normal, halfway, outgoing, incoming);
}
V3 eval_microfacet_reflection(V3 eta, V3 etak, float roughness, V3 normal, V3 outgoing,
    V3 incoming) {
  if (!same_side_up(normal, outgoing, incoming)) return {0, 0, 0};
  auto halfway = normalize(incoming + outgoing);
  return microfacet_reflection_with(fresnel_conductor(eta, etak, halfway, incoming), roughness,
      normal, halfway, outgoing, incoming);
}
V3 eval_microfacet_transmission(float ior, float roughness, V3 normal, V3 outgoing,
    V3 incoming) {  // math.h:4469-4483: reflection lobe mirrored through the surface, no Fresnel
  if (dot(normal, incoming) >= 0 || dot(normal, outgoing) <= 0) return {0, 0, 0};
  auto reflected = reflect(-incoming, normal);
  auto halfway   = normalize(reflected + outgoing);
  auto D         = microfacet_distribution(roughness, normal, halfway);
  auto G         = microfacet_shadowing(roughness, normal, halfway, outgoing, reflected);
  return V3{1, 1, 1} * D * G / (4 * dot(normal, outgoing) * dot(normal, reflected)) *
         (dot(normal, reflected));
}
struct RefractionSetup {  // the shared head of math.h:4486-4513 / 4610-4631
  bool  entering;
  V3    up_normal;
  float rel_ior;
};
RefractionSetup refraction_setup(float ior, V3 normal, V3 outgoing) {
  RefractionSetup s;
  s.entering  = dot(normal, outgoing) >= 0;
  s.up_normal = s.entering ? normal : -normal;
  s.rel_ior   = s.entering ? ior : (1 / ior);
  return s;
}
V3 eval_microfacet_refraction(float ior, float roughness, V3 normal, V3 outgoing, V3 incoming) {
  auto s = refraction_setup(ior, normal, outgoing);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    auto halfway = normalize(incoming + outgoing);
    auto F       = fresnel_dielectric(s.rel_ior, halfway, outgoing);
    auto D       = microfacet_distribution(roughness, s.up_normal, halfway);
    auto G       = microfacet_shadowing(roughness, s.up_normal, halfway, outgoing, incoming);
    return V3{1, 1, 1} * F * D * G / fabs_(4 * dot(normal, outgoing) * dot(normal, incoming)) *
           fabs_(dot(normal, incoming));
  } else {
    auto halfway = -normalize(s.rel_ior * incoming + outgoing) * (float)(s.entering ? 1 : -1);
    auto F       = fresnel_dielectric(s.rel_ior, halfway, outgoing);
    auto D       = microfacet_distribution(roughness, s.up_normal, halfway);
    auto G       = microfacet_shadowing(roughness, s.up_normal, halfway, outgoing, incoming);
    return V3{1, 1, 1} *
           fabs_((dot(outgoing, halfway) * dot(incoming, halfway)) /
                 (dot(outgoing, normal) * dot(incoming, normal))) *
           (1 - F) * D * G /
           std::pow(s.rel_ior * dot(halfway, incoming) + dot(halfway, outgoing), 2.0f) *
           fabs_(dot(normal, incoming));
  }
}
V3 sample_diffuse_reflection(V3 normal, V3 outgoing, float rx, float ry) {  // math.h:4515
  if (dot(normal, outgoing) <= 0) return {0, 0, 0};
  return sample_hemisphere_cos(normal, rx, ry);
}
// math.h:4528-4546: dielectric and conductor variants are the same code
V3 sample_microfacet_reflection(float roughness, V3 normal, V3 outgoing, float rx, float ry) {
  if (dot(normal, outgoing) <= 0) return {0, 0, 0};
  return reflect(outgoing, sample_microfacet(roughness, normal, rx, ry));
}
V3 sample_microfacet_transmission(float roughness, V3 normal, V3 outgoing, float rx, float ry) {
  if (dot(normal, outgoing) <= 0) return {0, 0, 0};  // math.h:4549-4556
  auto reflected = reflect(outgoing, sample_microfacet(roughness, normal, rx, ry));
  return -reflect(reflected, normal);
}
V3 sample_microfacet_refraction(float ior, float roughness, V3 normal, V3 outgoing, float rnl,
    float rx, float ry) {  // math.h:4559-4570
  auto s       = refraction_setup(ior, normal, outgoing);
  auto halfway = sample_microfacet(roughness, s.up_normal, rx, ry);
  if (rnl < fresnel_dielectric(s.entering ? ior : (1 / ior), halfway, outgoing))
    return reflect(outgoing, halfway);
  return refract(outgoing, halfway, s.entering ? (1 / ior) : ior);
}
float sample_diffuse_reflection_pdf(V3 normal, V3 outgoing, V3 incoming) {  // math.h:4572
  if (!same_side_up(normal, outgoing, incoming)) return 0;
  auto cosw = dot(normal, incoming);  // sample_hemisphere_cos_pdf, math.h:4874-4878
  return (cosw <= 0) ? 0 : cosw / pif;
}
float sample_microfacet_reflection_pdf(float roughness, V3 normal, V3 outgoing, V3 incoming) {
  if (!same_side_up(normal, outgoing, incoming)) return 0;  // math.h:4587-4607
  auto halfway = normalize(outgoing + incoming);
  return sample_microfacet_pdf(roughness, normal, halfway) / (4 * fabs_(dot(outgoing, halfway)));
}
float sample_microfacet_transmission_pdf(float roughness, V3 normal, V3 outgoing, V3 incoming) {
  if (dot(normal, incoming) >= 0 || dot(normal, outgoing) <= 0) return 0;  // math.h:4610-4619
  auto reflected = reflect(-incoming, normal);
  auto halfway   = normalize(reflected + outgoing);
  auto d         = sample_microfacet_pdf(roughness, normal, halfway);
  return d / (4 * fabs_(dot(outgoing, halfway)));
}
float sample_microfacet_refraction_pdf(float ior, float roughness, V3 normal, V3 outgoing,
    V3 incoming) {  // math.h:4622-4643
  auto s = refraction_setup(ior, normal, outgoing);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    auto halfway = normalize(incoming + outgoing);
    return fresnel_dielectric(s.rel_ior, halfway, outgoing) *
           sample_microfacet_pdf(roughness, s.up_normal, halfway) /
           (4 * fabs_(dot(outgoing, halfway)));
  } else {
    auto halfway = -normalize(s.rel_ior * incoming + outgoing) * (float)(s.entering ? 1 : -1);
    return (1 - fresnel_dielectric(s.rel_ior, halfway, outgoing)) *
           sample_microfacet_pdf(roughness, s.up_normal, halfway) *
           fabs_(dot(halfway, outgoing)) /
           std::pow(s.rel_ior * dot(halfway, incoming) + dot(halfway, outgoing), 2.0f);
  }
}

// --- delta lobes (math.h:4646-4755) -----------------------------------------
inline bool ior_is_one(float ior) { return fabs_(ior - 1) < 1e-3; }  // float vs double literal
V3 eval_delta_reflection(float ior, V3 normal, V3 outgoing, V3 incoming) {
  if (!same_side_up(normal, outgoing, incoming)) return {0, 0, 0};
  return V3{1, 1, 1} * fresnel_dielectric(ior, normal, outgoing);
}
V3 eval_delta_reflection(V3 eta, V3 etak, V3 normal, V3 outgoing, V3 incoming) {
  if (!same_side_up(normal, outgoing, incoming)) return {0, 0, 0};
  return fresnel_conductor(eta, etak, normal, outgoing);
}
V3 eval_delta_transmission(V3 normal, V3 outgoing, V3 incoming) {
  if (dot(normal, incoming) >= 0 || dot(normal, outgoing) <= 0) return {0, 0, 0};
  return {1, 1, 1};
}
V3 eval_delta_refraction(float ior, V3 normal, V3 outgoing, V3 incoming) {
  if (ior_is_one(ior))
    return dot(normal, incoming) * dot(normal, outgoing) <= 0 ? V3{1, 1, 1} : V3{0, 0, 0};
  auto s = refraction_setup(ior, normal, outgoing);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0)
    return V3{1, 1, 1} * fresnel_dielectric(s.rel_ior, s.up_normal, outgoing);
  return V3{1, 1, 1} * (1 / (s.rel_ior * s.rel_ior)) *
         (1 - fresnel_dielectric(s.rel_ior, s.up_normal, outgoing));
}
V3 sample_delta_reflection(V3 normal, V3 outgoing) {
  if (dot(normal, outgoing) <= 0) return {0, 0, 0};
  return reflect(outgoing, normal);
}
V3 sample_delta_transmission(V3 normal, V3 outgoing) {
  if (dot(normal, outgoing) <= 0) return {0, 0, 0};
  return -outgoing;
}
V3 sample_delta_refraction(float ior, V3 normal, V3 outgoing, float rnl) {
  if (ior_is_one(ior)) return -outgoing;
  auto s = refraction_setup(ior, normal, outgoing);
  if (rnl < fresnel_dielectric(s.rel_ior, s.up_normal, outgoing))
    return reflect(outgoing, s.up_normal);
  return refract(outgoing, s.up_normal, 1 / s.rel_ior);
}
float sample_delta_reflection_pdf(V3 normal, V3 outgoing, V3 incoming) {
  return same_side_up(normal, outgoing, incoming) ? 1 : 0;
}
float sample_delta_transmission_pdf(V3 normal, V3 outgoing, V3 incoming) {
  if (dot(normal, incoming) >= 0 || dot(normal, outgoing) <= 0) return 0;
  return 1;
}
float sample_delta_refraction_pdf(float ior, V3 normal, V3 outgoing, V3 incoming) {
  if (ior_is_one(ior)) return dot(normal, incoming) * dot(normal, outgoing) < 0 ? 1 : 0;
  auto s = refraction_setup(ior, normal, outgoing);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0)
    return fresnel_dielectric(s.rel_ior, s.up_normal, outgoing);
  return (1 - fresnel_dielectric(s.rel_ior, s.up_normal, outgoing));
}

// ---------------------------------------------------------------------------
// eval_brdf and the lobe dispatch (pt.cpp:372-394, 405-495, 1069-1280).
// Textures are out of scope: every eval_texture(nullptr) factor is 1.
// ---------------------------------------------------------------------------
struct Brdf {
  V3       diffuse{0, 0, 0}, specular{0, 0, 0}, metal{0, 0, 0}, transmission{0, 0, 0},
      refraction{0, 0, 0};
  float    roughness = 0, opacity = 1, ior = 1;
  V3       meta{0, 0, 0}, metak{0, 0, 0};
  float    diffuse_pdf = 0, specular_pdf = 0, metal_pdf = 0, transmission_pdf = 0,
        refraction_pdf = 0;
  bool     hair = false;
  HairBrdf hair_brdf;
};
// color_tex = eval_texture(color_tex, texcoord, false); emission_tex_x = eval_texture(emission_tex,
// texcoord, true).x — the reference multiplies TRANSMISSION by the emission texture (pt.cpp:421-422)
Brdf surface_brdf(const Material& mat, V3 normal, V3 outgoing, V3 color_tex = {1, 1, 1}, float emission_tex_x = 1.0f) {  // pt.cpp:405-471
  auto base         = mat.color * color_tex;
  auto specular     = mat.specular * 1.0f;
  auto metallic     = mat.metallic * 1.0f;
  auto roughness    = mat.roughness * 1.0f;
  auto ior          = mat.ior;
  auto transmission = mat.transmission * emission_tex_x;
  auto opacity      = mat.opacity * ((1.0f + 1.0f + 1.0f) / 3);
  auto thin         = mat.thin || !mat.transmission;

  Brdf brdf;
  auto weight = V3{1, 1, 1};
  brdf.metal  = weight * metallic;
  weight      = weight * (1 - metallic);
  brdf.refraction = thin ? V3{0, 0, 0} : (weight * transmission);
  weight          = weight * (1 - (thin ? 0 : transmission));
  brdf.specular   = weight * specular;
  weight          = weight * (1 - specular * fresnel_dielectric(ior, outgoing, normal));
  brdf.transmission = thin ? (weight * transmission * base) : V3{0, 0, 0};
  weight            = weight * (1 - (thin ? transmission : 0));
  brdf.diffuse   = weight * base;
  brdf.meta      = reflectivity_to_eta(base);
  brdf.metak     = {0, 0, 0};
  brdf.roughness = roughness * roughness;
  brdf.ior       = ior;
  brdf.opacity   = opacity;
  if (nonzero(brdf.diffuse) || brdf.roughness)  // `!=` on vectors is !(==), math.h:1999
    brdf.roughness = fclamp(brdf.roughness, 0.03f * 0.03f, 1.0f);
  if (!nonzero(brdf.specular) && !nonzero(brdf.metal) && !nonzero(brdf.transmission) &&
      !nonzero(brdf.refraction))
    brdf.roughness = 1;
  if (brdf.opacity > 0.999f) brdf.opacity = 1;

  brdf.diffuse_pdf  = hmax(brdf.diffuse);
  brdf.specular_pdf = hmax(brdf.specular * fresnel_dielectric(brdf.ior, normal, outgoing));
  brdf.metal_pdf    = hmax(brdf.metal * fresnel_conductor(brdf.meta, brdf.metak, normal, outgoing));
  brdf.transmission_pdf = hmax(brdf.transmission);
  brdf.refraction_pdf   = hmax(brdf.refraction);
  auto pdf_sum = brdf.diffuse_pdf + brdf.specular_pdf + brdf.metal_pdf + brdf.transmission_pdf +
                 brdf.refraction_pdf;
  if (pdf_sum) {
    brdf.diffuse_pdf /= pdf_sum, brdf.specular_pdf /= pdf_sum, brdf.metal_pdf /= pdf_sum;
    brdf.transmission_pdf /= pdf_sum, brdf.refraction_pdf /= pdf_sum;
  }
  return brdf;
}
Brdf eval_brdf(const yo_scene& scene, int object, int element, const float uv[2],
    V3 normal, V3 outgoing) {
  auto& obj   = scene.objects[object];
  auto& mat   = scene.materials[obj.material];
  auto& shape = scene.shapes[obj.shape];
  float tc[2];
  eval_texcoord(scene, object, element, uv, tc);
  auto brdf = surface_brdf(mat, normal, outgoing, eval_texture(texture_of(scene, mat.color_tex), tc, false),
      eval_texture(texture_of(scene, mat.emission_tex), tc, true).x);
  brdf.hair   = shape.nlines() > 0;  // pt.cpp:474
  if (brdf.hair) {
    auto tangent   = eval_normal(scene, object, element, uv);
    brdf.hair_brdf = eval_hair_brdf(mat.hair, uv[1], normal, tangent);
  }
  return brdf;
}
inline bool is_delta(const Brdf& brdf) { return !brdf.roughness; }  // pt.cpp:495

V3 eval_brdfcos(const Brdf& brdf, V3 normal, V3 outgoing, V3 incoming) {  // pt.cpp:1069-1102
  if (brdf.hair) return eval_hair_scattering(brdf.hair_brdf, outgoing, incoming);
  if (!brdf.roughness) return {0, 0, 0};
  auto brdfcos = V3{0, 0, 0};
  if (nonzero(brdf.diffuse))
    brdfcos = brdfcos + brdf.diffuse * eval_diffuse_reflection(normal, outgoing, incoming);
  if (nonzero(brdf.specular))
    brdfcos = brdfcos + brdf.specular * eval_microfacet_reflection(brdf.ior, brdf.roughness,
                                            normal, outgoing, incoming);
  if (nonzero(brdf.metal))
    brdfcos = brdfcos + brdf.metal * eval_microfacet_reflection(brdf.meta, brdf.metak,
                                         brdf.roughness, normal, outgoing, incoming);
  if (nonzero(brdf.transmission))
    brdfcos = brdfcos + brdf.transmission * eval_microfacet_transmission(brdf.ior,
                                                brdf.roughness, normal, outgoing, incoming);
  if (nonzero(brdf.refraction))
    brdfcos = brdfcos + brdf.refraction * eval_microfacet_refraction(brdf.ior, brdf.roughness,
                                              normal, outgoing, incoming);
  return brdfcos;
}
V3 eval_delta(const Brdf& brdf, V3 normal, V3 outgoing, V3 incoming) {  // pt.cpp:1104-1128
  if (brdf.roughness) return {0, 0, 0};
  auto brdfcos = V3{0, 0, 0};
  if (nonzero(brdf.specular) && !nonzero(brdf.refraction))
    brdfcos = brdfcos + brdf.specular * eval_delta_reflection(brdf.ior, normal, outgoing, incoming);
  if (nonzero(brdf.metal))
    brdfcos = brdfcos + brdf.metal * eval_delta_reflection(brdf.meta, brdf.metak, normal,
                                         outgoing, incoming);
  if (nonzero(brdf.transmission))
    brdfcos = brdfcos + brdf.transmission * eval_delta_transmission(normal, outgoing, incoming);
  if (nonzero(brdf.refraction))
    brdfcos = brdfcos + brdf.refraction * eval_delta_refraction(brdf.ior, normal, outgoing, incoming);
  return brdfcos;
}
V3 sample_brdfcos(const Brdf& brdf, V3 normal, V3 outgoing, float rnl, float rnx,
    float rny) {  // pt.cpp:1131-1175
  if (brdf.hair) return sample_hair_scattering(brdf.hair_brdf, outgoing, rnx, rny);
  if (!brdf.roughness) return {0, 0, 0};
  auto cdf = 0.0f;
  if (brdf.diffuse_pdf) {
    cdf += brdf.diffuse_pdf;
    if (rnl < cdf) return sample_diffuse_reflection(normal, outgoing, rnx, rny);
  }
  if (brdf.specular_pdf && !brdf.refraction_pdf) {
    cdf += brdf.specular_pdf;
    if (rnl < cdf) return sample_microfacet_reflection(brdf.roughness, normal, outgoing, rnx, rny);
  }
  if (brdf.metal_pdf) {
    cdf += brdf.metal_pdf;
    if (rnl < cdf) return sample_microfacet_reflection(brdf.roughness, normal, outgoing, rnx, rny);
  }
  if (brdf.transmission_pdf) {
    cdf += brdf.transmission_pdf;
    if (rnl < cdf) return sample_microfacet_transmission(brdf.roughness, normal, outgoing, rnx, rny);
  }
  if (brdf.refraction_pdf) {
    cdf += brdf.refraction_pdf;
    if (rnl < cdf)
      return sample_microfacet_refraction(brdf.ior, brdf.roughness, normal, outgoing, rnl, rnx, rny);
  }
  return {0, 0, 0};
}
V3 sample_delta(const Brdf& brdf, V3 normal, V3 outgoing, float rnl) {  // pt.cpp:1177-1214
  if (brdf.roughness) return {0, 0, 0};
  auto cdf = 0.0f;
  cdf += brdf.diffuse_pdf;
  if (brdf.specular_pdf && !brdf.refraction_pdf) {
    cdf += brdf.specular_pdf;
    if (rnl < cdf) return sample_delta_reflection(normal, outgoing);
  }
  if (brdf.metal_pdf) {
    cdf += brdf.metal_pdf;
    if (rnl < cdf) return sample_delta_reflection(normal, outgoing);
  }
  if (brdf.transmission_pdf) {
    cdf += brdf.transmission_pdf;
    if (rnl < cdf) return sample_delta_transmission(normal, outgoing);
  }
  if (brdf.refraction_pdf) {
    cdf += brdf.refraction_pdf;
    if (rnl < cdf) return sample_delta_refraction(brdf.ior, normal, outgoing, rnl);
  }
  return {0, 0, 0};
}
float sample_brdfcos_pdf(const Brdf& brdf, V3 normal, V3 outgoing, V3 incoming) {
  if (brdf.hair)  // pt.cpp:1217-1256
    return sample_hair_scattering_pdf(brdf.hair_brdf, outgoing, incoming);
  if (!brdf.roughness) return 0;
  auto pdf = 0.0f;
  if (brdf.diffuse_pdf)
    pdf += brdf.diffuse_pdf * sample_diffuse_reflection_pdf(normal, outgoing, incoming);
  if (brdf.specular_pdf && !brdf.refraction_pdf)
    pdf += brdf.specular_pdf *
           sample_microfacet_reflection_pdf(brdf.roughness, normal, outgoing, incoming);
  if (brdf.metal_pdf)
    pdf += brdf.metal_pdf *
           sample_microfacet_reflection_pdf(brdf.roughness, normal, outgoing, incoming);
  if (brdf.transmission_pdf)
    pdf += brdf.transmission_pdf *
           sample_microfacet_transmission_pdf(brdf.roughness, normal, outgoing, incoming);
  if (brdf.refraction_pdf)
    pdf += brdf.refraction_pdf * sample_microfacet_refraction_pdf(brdf.ior, brdf.roughness,
                                     normal, outgoing, incoming);
  return pdf;
}
float sample_delta_pdf(const Brdf& brdf, V3 normal, V3 outgoing, V3 incoming) {
  if (brdf.roughness) return 0;  // pt.cpp:1258-1280
  auto pdf = 0.0f;
  if (brdf.specular_pdf && !brdf.refraction_pdf)
    pdf += brdf.specular_pdf * sample_delta_reflection_pdf(normal, outgoing, incoming);
  if (brdf.metal_pdf)
    pdf += brdf.metal_pdf * sample_delta_reflection_pdf(normal, outgoing, incoming);
  if (brdf.transmission_pdf)
    pdf += brdf.transmission_pdf * sample_delta_transmission_pdf(normal, outgoing, incoming);
  if (brdf.refraction_pdf)
    pdf += brdf.refraction_pdf * sample_delta_refraction_pdf(brdf.ior, normal, outgoing, incoming);
  return pdf;
}

// ---------------------------------------------------------------------------
// Environment and lights (pt.cpp:148-200, 536-547, 1283-1358, 1695-1740)
// ---------------------------------------------------------------------------
V3 eval_texture(const Environment& env, float u_, float v_) {  // pt.cpp:167-200
  if (env.texels.empty()) return {1, 1, 1};
  auto sx = env.w, sy = env.h;
  auto s = std::fmod(u_, 1.0f) * sx;
  if (s < 0) s += sx;
  auto t = std::fmod(v_, 1.0f) * sy;
  if (t < 0) t += sy;
  auto i = iclamp((int)s, 0, sx - 1), j = iclamp((int)t, 0, sy - 1);
  auto ii = (i + 1) % sx, jj = (j + 1) % sy;
  auto u = s - i, v = t - j;
  auto px = [&](int a, int b) { return env.texels[(size_t)b * sx + a]; };
  return px(i, j) * (1 - u) * (1 - v) + px(i, jj) * (1 - u) * v +
         px(ii, j) * u * (1 - v) + px(ii, jj) * u * v;
}
V3 eval_environment(const yo_scene& scene, V3 dir) {  // pt.cpp:536-547
  auto emission = V3{0, 0, 0};
  for (auto& env : scene.environments) {
    auto wl = transform_direction(inverse(env.frame, false), dir);
    auto tx = std::atan2(wl.z, wl.x) / (2 * pif);
    auto ty = std::acos(fclamp(wl.y, -1.0f, 1.0f)) / pif;
    if (tx < 0) tx += 1;
    tls_counters.envl++;
    emission = emission + env.emission * eval_texture(env, tx, ty);
  }
  return emission;
}
V3 sample_lights(const yo_scene& scene, V3 position, float rl, float rel,
    float ruvx, float ruvy) {  // pt.cpp:1283-1308
  auto  n        = (int)scene.lights.size();
  auto  light_id = iclamp((int)(rl * n), 0, n - 1);  // math.h:4927-4929
  auto& light    = scene.lights[light_id];
  if (light.object >= 0) {
    auto  element = sample_discrete_cdf(light.cdf, rel);
    float uv[2]   = {1 - std::sqrt(ruvx), ruvy * std::sqrt(ruvx)};  // math.h:4910
    auto  lposition = eval_position(scene, light.object, element, uv);
    return normalize(lposition - position);
  } else if (light.environment >= 0) {
    auto& env = scene.environments[light.environment];
    if (!env.texels.empty()) {
      tls_counters.envs++;
      auto idx = sample_discrete_cdf(light.cdf, rel);
      auto ux  = (idx % env.w + 0.5f) / env.w;
      auto uy  = (idx / env.w + 0.5f) / env.h;
      return transform_direction(env.frame,
          {std::cos(ux * 2 * pif) * std::sin(uy * pif), std::cos(uy * pif),
              std::sin(ux * 2 * pif) * std::sin(uy * pif)});
    } else {
      return sample_sphere(ruvx, ruvy);
    }
  }
  return {0, 0, 0};
}
float sample_lights_pdf(const yo_scene& scene, V3 position, V3 direction) {
  auto pdf = 0.0f;  // pt.cpp:1311-1358
  for (auto& light : scene.lights) {
    if (light.object >= 0) {
      auto lpdf          = 0.0f;
      auto next_position = position;
      for (auto bounce = 0; bounce < 100; bounce++) {
        int   element = -1;
        float uv[2]   = {0, 0};
        float dist    = 0;
        Ray   ray{next_position, direction};
        if (!intersect_instance_bvh(scene, light.object, ray, element, uv, dist))
          break;
        auto lposition = eval_position(scene, light.object, element, uv);
        auto lnormal   = eval_element_normal(scene, light.object, element);
        auto area      = light.cdf.back();
        auto dp        = lposition - position;
        lpdf += dot(dp, dp) / (fabs_(dot(lnormal, direction)) * area);
        next_position = lposition + direction * 1e-3f;
      }
      pdf += lpdf;
    } else if (light.environment >= 0) {
      auto& env = scene.environments[light.environment];
      if (!env.texels.empty()) {
        auto wl = transform_direction(inverse(env.frame, false), direction);
        auto tx = std::atan2(wl.z, wl.x) / (2 * pif);
        auto ty = std::acos(fclamp(wl.y, -1.0f, 1.0f)) / pif;
        if (tx < 0) tx += 1;
        auto i    = iclamp((int)(tx * env.w), 0, env.w - 1);
        auto j    = iclamp((int)(ty * env.h), 0, env.h - 1);
        auto prob = sample_discrete_cdf_pdf(light.cdf, j * env.w + i) / light.cdf.back();
        auto angle = (2 * pif / env.w) * (pif / env.h) *
                     std::sin(pif * (j + 0.5f) / env.h);
        pdf += prob / angle;
      } else {
        pdf += 1 / (4 * pif);
      }
    }
  }
  pdf *= (float)1 / (float)(int)scene.lights.size();  // math.h:4930
  return pdf;
}
void init_lights(yo_scene& scene) {  // pt.cpp:1695-1740
  scene.lights.clear();
  for (auto oid = 0; oid < (int)scene.objects.size(); oid++) {
    auto& obj = scene.objects[oid];
    if (scene.materials[obj.material].emission == V3{0, 0, 0}) continue;
    auto& shape = scene.shapes[obj.shape];
    if (!shape.ntriangles()) continue;
    Light light;
    light.object = oid;
    light.cdf.resize(shape.ntriangles());
    for (auto idx = 0; idx < (int)light.cdf.size(); idx++) {
      auto p0 = shape.positions[shape.triangles[3 * idx]];
      auto p1 = shape.positions[shape.triangles[3 * idx + 1]];
      auto p2 = shape.positions[shape.triangles[3 * idx + 2]];
      light.cdf[idx] = length(cross(p1 - p0, p2 - p0)) / 2;  // math.h:3306
      if (idx) light.cdf[idx] += light.cdf[idx - 1];
    }
    scene.lights.push_back(std::move(light));
  }
  for (auto eid = 0; eid < (int)scene.environments.size(); eid++) {
    auto& env = scene.environments[eid];
    if (env.emission == V3{0, 0, 0}) continue;
    Light light;
    light.environment = eid;
    if (!env.texels.empty()) {
      light.cdf.resize((size_t)env.w * env.h);
      for (auto i = 0; i < (int)light.cdf.size(); i++) {
        auto iy      = i / env.w;
        auto th      = (iy + 0.5f) * pif / env.h;
        auto value   = env.texels[i];
        light.cdf[i] = hmax(value) * std::sin(th);
        if (i) light.cdf[i] += light.cdf[i - 1];
      }
    }
    scene.lights.push_back(std::move(light));
  }
}

// ---------------------------------------------------------------------------
// Camera and path tracing (pt.cpp:211-229, 1380-1511, 1676-1689)
// ---------------------------------------------------------------------------
Ray sample_camera(const Camera& cam, int i, int j, int w, int h, float pu, float pv,
    float lu, float lv) {
  auto uvx = (i + pu) / w, uvy = (j + pv) / h;  // pt.cpp:228
  auto r   = std::sqrt(lv);                     // sample_disk, math.h:4895
  auto phi = 2 * pif * lu;
  auto lx = std::cos(phi) * r, ly = std::sin(phi) * r;
  auto q  = V3{cam.film[0] * (0.5f - uvx), cam.film[1] * (uvy - 0.5f), cam.lens};
  auto dc = -normalize(q);
  auto e  = V3{lx * cam.aperture / 2, ly * cam.aperture / 2, 0};
  auto p  = dc * cam.focus / fabs_(dc.z);
  auto d  = normalize(p - e);
  return {transform_point(cam.frame, e), transform_direction(cam.frame, d)};
}

struct Vec4 {
  float x, y, z, w;
};
// ---------------------------------------------------------------------------
// Homogeneous volumes (pt.cpp:498-533, 1360-1377; math.h:4758-4822)
// ---------------------------------------------------------------------------
struct Vsdf {
  V3    density{0, 0, 0}, scatter{0, 0, 0};
  float anisotropy = 0;
};
inline bool has_volume(const Material& mat) { return !mat.thin && mat.transmission; }  // pt.cpp:531
Vsdf eval_vsdf(const Material& mat, V3 color_tex, float emission_tex_x, V3 scattering_tex) {  // pt.cpp:504-527
  auto base         = mat.color * color_tex;
  auto transmission = mat.transmission * emission_tex_x;
  auto thin         = mat.thin || !mat.transmission;
  Vsdf v;
  if (transmission && !thin) {
    auto c    = V3{fclamp(base.x, 0.0001f, 1.0f), fclamp(base.y, 0.0001f, 1.0f), fclamp(base.z, 0.0001f, 1.0f)};
    v.density = -vlog(c) / mat.trdepth;
  }
  v.scatter    = mat.scattering * scattering_tex;
  v.anisotropy = mat.scanisotropy;
  return v;
}
inline V3 eval_transmittance(V3 density, float distance) { return vexp(-density * distance); }
float sample_transmittance(V3 density, float max_distance, float rl, float rd) {
  auto channel  = iclamp((int)(rl * 3), 0, 2);
  auto distance = (at(density, channel) == 0) ? flt_max : -std::log(1 - rd) / at(density, channel);
  return fmin_(distance, max_distance);
}
float sample_transmittance_pdf(V3 density, float distance, float max_distance) {
  auto sum3 = [](V3 a) { return a.x + a.y + a.z; };
  if (distance < max_distance) return sum3(density * vexp(-density * distance)) / 3;
  return sum3(vexp(-density * max_distance)) / 3;
}
float eval_phasefunction(float anisotropy, V3 outgoing, V3 incoming) {  // Henyey-Greenstein
  auto cosine = -dot(outgoing, incoming);
  auto denom  = 1 + anisotropy * anisotropy - 2 * anisotropy * cosine;
  return (1 - anisotropy * anisotropy) / (4 * pif * denom * std::sqrt(denom));
}
V3 sample_phasefunction(float anisotropy, V3 outgoing, float rx, float ry) {
  auto cos_theta = 0.0f;
  if (fabs_(anisotropy) < 1e-3f) {
    cos_theta = 1 - 2 * ry;
  } else {
    float square = (1 - anisotropy * anisotropy) / (1 + anisotropy - 2 * anisotropy * ry);
    cos_theta    = (1 + anisotropy * anisotropy - square * square) / (2 * anisotropy);
  }
  auto sin_theta = std::sqrt(fmax_(0.0f, 1 - cos_theta * cos_theta));
  auto phi       = 2 * pif * rx;
  auto local     = V3{sin_theta * std::cos(phi), sin_theta * std::sin(phi), cos_theta};
  // basis_fromz(-outgoing) * local: a matrix product, not normalised (math.h:4815)
  auto zz   = normalize(-outgoing);
  auto sign = copysignf(1.0f, zz.z);
  auto a    = -1.0f / (sign + zz.z);
  auto b    = zz.x * zz.y * a;
  auto x    = V3{1.0f + sign * zz.x * zz.x * a, sign * b, -sign * zz.x};
  auto y    = V3{b, sign + zz.y * zz.y * a, -zz.y};
  return x * local.x + y * local.y + zz * local.z;
}
inline V3 eval_scattering(const Vsdf& v, V3 outgoing, V3 incoming) {  // pt.cpp:1360-1365
  if (v.density == V3{0, 0, 0}) return {0, 0, 0};
  return v.scatter * v.density * eval_phasefunction(v.anisotropy, outgoing, incoming);
}
inline V3 sample_scattering(const Vsdf& v, V3 outgoing, float rx, float ry) {
  if (v.density == V3{0, 0, 0}) return {0, 0, 0};
  return sample_phasefunction(v.anisotropy, outgoing, rx, ry);
}
inline float sample_scattering_pdf(const Vsdf& v, V3 outgoing, V3 incoming) {
  if (v.density == V3{0, 0, 0}) return 0;
  return eval_phasefunction(v.anisotropy, outgoing, incoming);
}

Vec4 trace_path(const yo_scene& scene, const Ray& ray_, Rng& rng, int bounces) {
  auto radiance = V3{0, 0, 0};
  auto weight   = V3{1, 1, 1};
  auto ray      = ray_;
  auto hit      = false;
  std::vector<Vsdf> volume_stack;
  for (auto bounce = 0; bounce < bounces; bounce++) {
    int   object = -1, element = -1;
    float uv[2] = {0, 0}, distance = 0;
    if (!intersect_scene_bvh(scene, ray, object, element, uv, distance)) {
      radiance = radiance + weight * eval_environment(scene, ray.d);
      break;
    }
    // inside a volume: sample the free-flight distance (pt.cpp:1403-1414). g++ evaluates
    // the two rand1f arguments right to left: rd is drawn before rl.
    auto in_volume = false;
    if (!volume_stack.empty()) {
      auto& vsdf = volume_stack.back();
      auto  rd = rand1f(rng), rl = rand1f(rng);
      auto  dist = sample_transmittance(vsdf.density, distance, rl, rd);
      weight     = weight * (eval_transmittance(vsdf.density, dist) /
                            sample_transmittance_pdf(vsdf.density, dist, distance));
      in_volume  = dist < distance;
      distance   = dist;
    }
    if (in_volume) {  // pt.cpp:1472-1497: scatter inside the medium
      auto  outgoing = -ray.d;
      auto  position = ray.o + ray.d * distance;
      auto& vsdf     = volume_stack.back();
      hit           = true;
      auto incoming = V3{0, 0, 0};
      if (rand1f(rng) < 0.5f) {
        auto rnx = rand1f(rng), rny = rand1f(rng);
        auto rnl = rand1f(rng);
        (void)rnl;
        incoming = sample_scattering(vsdf, outgoing, rnx, rny);
      } else {
        auto ruvx = rand1f(rng), ruvy = rand1f(rng);
        auto rel = rand1f(rng);
        auto rl  = rand1f(rng);
        incoming = sample_lights(scene, position, rl, rel, ruvx, ruvy);
      }
      weight = weight * (eval_scattering(vsdf, outgoing, incoming) /
                            (0.5f * sample_scattering_pdf(vsdf, outgoing, incoming) +
                                0.5f * sample_lights_pdf(scene, position, incoming)));
      ray = Ray{position, incoming};
      if (weight == V3{0, 0, 0} || !finite3(weight)) break;
      if (bounce > 3) {
        auto rr_prob = fmin_((float)0.99, hmax(weight));
        if (rand1f(rng) >= rr_prob) break;
        weight = weight * (1 / rr_prob);
      }
      continue;
    }
    auto outgoing = -ray.d;
    auto position = eval_position(scene, object, element, uv);
    auto normal   = eval_shading_normal(scene, object, element, uv, outgoing);
    auto& hit_mat = scene.materials[scene.objects[object].material];
    float tc[2];
    eval_texcoord(scene, object, element, uv, tc);
    auto emission = hit_mat.emission * eval_texture(texture_of(scene, hit_mat.emission_tex), tc);  // pt.cpp:397-402
    auto brdf     = eval_brdf(scene, object, element, uv, normal, outgoing);
    if (brdf.hair) tls_counters.hair++; else tls_counters.surf++;
    if (brdf.opacity < 1 && rand1f(rng) >= brdf.opacity) {  // pt.cpp:1429-1433
      ray = Ray{position + ray.d * 1e-2f, ray.d};
      bounce -= 1;
      continue;
    }
    hit      = true;
    radiance = radiance + weight * emission;
    auto incoming = V3{0, 0, 0};
    if (!is_delta(brdf)) {
      if (rand1f(rng) < 0.5f) {
        // g++ evaluates call arguments right to left: rn (x then y), then rnl
        auto rnx = rand1f(rng), rny = rand1f(rng);
        auto rnl = rand1f(rng);
        incoming = sample_brdfcos(brdf, normal, outgoing, rnl, rnx, rny);
      } else {
        // ruv (x then y), then rel, then rl
        auto ruvx = rand1f(rng), ruvy = rand1f(rng);
        auto rel = rand1f(rng);
        auto rl  = rand1f(rng);
        incoming = sample_lights(scene, position, rl, rel, ruvx, ruvy);
      }
      weight = weight * (eval_brdfcos(brdf, normal, outgoing, incoming) /
                            (0.5f * sample_brdfcos_pdf(brdf, normal, outgoing, incoming) +
                                0.5f * sample_lights_pdf(scene, position, incoming)));
    } else {  // pt.cpp:1452-1456
      incoming = sample_delta(brdf, normal, outgoing, rand1f(rng));
      weight   = weight * (eval_delta(brdf, normal, outgoing, incoming) /
                            sample_delta_pdf(brdf, normal, outgoing, incoming));
    }
    // entering / leaving a closed transmissive object (pt.cpp:1458-1467)
    if (has_volume(scene.materials[scene.objects[object].material]) &&
        dot(normal, outgoing) * dot(normal, incoming) < 0) {
      if (volume_stack.empty())
        volume_stack.push_back(eval_vsdf(hit_mat, eval_texture(texture_of(scene, hit_mat.color_tex), tc, false),
            eval_texture(texture_of(scene, hit_mat.emission_tex), tc, true).x,
            eval_texture(texture_of(scene, hit_mat.scattering_tex), tc, false)));
      else volume_stack.pop_back();
    }
    ray = Ray{position, incoming};
    if (weight == V3{0, 0, 0} || !finite3(weight)) break;
    if (bounce > 3) {
      auto rr_prob = fmin_((float)0.99, hmax(weight));
      if (rand1f(rng) >= rr_prob) break;
      weight = weight * (1 / rr_prob);
    }
  }
  return {radiance.x, radiance.y, radiance.z, hit ? 1.0f : 0.0f};
}

// trace_naive (pt.cpp:1514-1581): brdf sampling only (no light sampling, no MIS), no volumes.
Vec4 trace_naive(const yo_scene& scene, const Ray& ray_, Rng& rng, int bounces) {
  auto radiance = V3{0, 0, 0};
  auto weight   = V3{1, 1, 1};
  auto ray      = ray_;
  auto hit      = false;
  for (auto bounce = 0; bounce < bounces; bounce++) {
    int   object = -1, element = -1;
    float uv[2] = {0, 0}, distance = 0;
    if (!intersect_scene_bvh(scene, ray, object, element, uv, distance)) {
      radiance = radiance + weight * eval_environment(scene, ray.d);
      break;
    }
    auto outgoing = -ray.d;
    auto position = eval_position(scene, object, element, uv);
    auto normal   = eval_shading_normal(scene, object, element, uv, outgoing);
    auto& hit_mat = scene.materials[scene.objects[object].material];
    float tc[2];
    eval_texcoord(scene, object, element, uv, tc);
    auto emission = hit_mat.emission * eval_texture(texture_of(scene, hit_mat.emission_tex), tc);
    auto brdf     = eval_brdf(scene, object, element, uv, normal, outgoing);
    if (brdf.hair) tls_counters.hair++; else tls_counters.surf++;
    if (brdf.opacity < 1 && rand1f(rng) >= brdf.opacity) {
      ray = Ray{position + ray.d * 1e-2f, ray.d};
      bounce -= 1;
      continue;
    }
    hit      = true;
    radiance = radiance + weight * emission;
    auto incoming = V3{0, 0, 0};
    if (!is_delta(brdf)) {
      // sample_brdfcos(brdf, normal, outgoing, rand1f(rng), rand2f(rng)): g++ draws rn first
      auto rnx = rand1f(rng), rny = rand1f(rng);
      auto rnl = rand1f(rng);
      incoming = sample_brdfcos(brdf, normal, outgoing, rnl, rnx, rny);
      weight   = weight * (eval_brdfcos(brdf, normal, outgoing, incoming) /
                            sample_brdfcos_pdf(brdf, normal, outgoing, incoming));
    } else {
      incoming = sample_delta(brdf, normal, outgoing, rand1f(rng));
      weight   = weight * (eval_delta(brdf, normal, outgoing, incoming) /
                            sample_delta_pdf(brdf, normal, outgoing, incoming));
    }
    if (weight == V3{0, 0, 0} || !finite3(weight)) break;
    if (bounce > 3) {
      auto rr_prob = fmin_((float)0.99, hmax(weight));
      if (rand1f(rng) >= rr_prob) break;
      weight = weight * (1 / rr_prob);
    }
    ray = Ray{position, incoming};
  }
  return {radiance.x, radiance.y, radiance.z, hit ? 1.0f : 0.0f};
}

// trace_eyelight (pt.cpp:1584-1641): the light sits at the eye; only delta chains continue.
Vec4 trace_eyelight(const yo_scene& scene, const Ray& ray_, Rng& rng, int bounces) {
  auto radiance = V3{0, 0, 0};
  auto weight   = V3{1, 1, 1};
  auto ray      = ray_;
  auto hit      = false;
  for (auto bounce = 0; bounce < (bounces > 4 ? bounces : 4); bounce++) {
    int   object = -1, element = -1;
    float uv[2] = {0, 0}, distance = 0;
    if (!intersect_scene_bvh(scene, ray, object, element, uv, distance)) {
      radiance = radiance + weight * eval_environment(scene, ray.d);
      break;
    }
    auto outgoing = -ray.d;
    auto position = eval_position(scene, object, element, uv);
    auto normal   = eval_shading_normal(scene, object, element, uv, outgoing);
    auto& hit_mat = scene.materials[scene.objects[object].material];
    float tc[2];
    eval_texcoord(scene, object, element, uv, tc);
    auto emission = hit_mat.emission * eval_texture(texture_of(scene, hit_mat.emission_tex), tc);
    auto brdf     = eval_brdf(scene, object, element, uv, normal, outgoing);
    if (brdf.hair) tls_counters.hair++; else tls_counters.surf++;
    if (brdf.opacity < 1 && rand1f(rng) >= brdf.opacity) {
      ray = Ray{position + ray.d * 1e-2f, ray.d};
      bounce -= 1;
      continue;
    }
    hit      = true;
    radiance = radiance + weight * emission;
    auto incoming = outgoing;
    radiance = radiance + weight * pif * eval_brdfcos(brdf, normal, outgoing, incoming);
    if (!is_delta(brdf)) break;
    incoming = sample_delta(brdf, normal, outgoing, rand1f(rng));
    weight   = weight * (eval_delta(brdf, normal, outgoing, incoming) /
                          sample_delta_pdf(brdf, normal, outgoing, incoming));
    if (weight == V3{0, 0, 0} || !finite3(weight)) break;
    ray = Ray{position, incoming};
  }
  return {radiance.x, radiance.y, radiance.z, hit ? 1.0f : 0.0f};
}

// trace_normal (pt.cpp:1644-1658): alpha is 1 for hits and misses alike.
Vec4 trace_normal(const yo_scene& scene, const Ray& ray, Rng&, int) {
  int   object = -1, element = -1;
  float uv[2] = {0, 0}, distance = 0;
  if (!intersect_scene_bvh(scene, ray, object, element, uv, distance)) {
    auto e = eval_environment(scene, ray.d);
    return {e.x, e.y, e.z, 1};
  }
  auto normal = eval_shading_normal(scene, object, element, uv, -ray.d);
  auto c      = normal * 0.5f + 0.5f;
  return {c.x, c.y, c.z, 1};
}

struct Pixel {  // pt.h:419-423
  Vec4 accumulated = {0, 0, 0, 0};
  int  samples     = 0;
  Rng  rng;
};
Vec4 trace_sample(const yo_scene& scene, Pixel& pixel, int i, int j, int w, int h,
    int bounces, float clamp, int shader = YH_SHADER_PATH) {  // pt.cpp:1676-1689
  // argument order under g++: lens uv first, then pixel uv
  auto lu = rand1f(pixel.rng), lv = rand1f(pixel.rng);
  auto pu = rand1f(pixel.rng), pv = rand1f(pixel.rng);
  auto ray    = sample_camera(scene.camera, i, j, w, h, pu, pv, lu, lv);
  // get_trace_shader_func (pt.cpp:1660-1672)
  auto shaded = shader == YH_SHADER_NAIVE      ? trace_naive(scene, ray, pixel.rng, bounces)
                : shader == YH_SHADER_EYELIGHT ? trace_eyelight(scene, ray, pixel.rng, bounces)
                : shader == YH_SHADER_NORMAL   ? trace_normal(scene, ray, pixel.rng, bounces)
                                               : trace_path(scene, ray, pixel.rng, bounces);
  auto rgb    = V3{shaded.x, shaded.y, shaded.z};
  if (!finite3(rgb)) rgb = {0, 0, 0};
  if (hmax(rgb) > clamp) rgb = rgb * (clamp / hmax(rgb));
  pixel.accumulated.x += rgb.x, pixel.accumulated.y += rgb.y;
  pixel.accumulated.z += rgb.z, pixel.accumulated.w += shaded.w;
  pixel.samples += 1;
  tls_counters.samples++;
  auto n = (float)pixel.samples;
  return {pixel.accumulated.x / n, pixel.accumulated.y / n, pixel.accumulated.z / n,
      pixel.accumulated.w / n};
}

void image_size(const Camera& cam, int resolution, int& w, int& h) {
  if (cam.film[0] > cam.film[1]) {  // pt.cpp:1933-1939
    w = resolution;
    h = (int)round(resolution * cam.film[1] / cam.film[0]);
  } else {
    w = (int)round(resolution * cam.film[0] / cam.film[1]);
    h = resolution;
  }
}

HairMaterial mkhair(const float* m) {
  HairMaterial h;
  std::memcpy(&h, m, 48);
  return h;
}
HairBrdf mkbrdf(const float* p) {
  HairBrdf b;
  std::memcpy(&b, p, 120);
  return b;
}
V3  v3(const float* p) { return {p[0], p[1], p[2]}; }
Ray mkray(const float* r) { return {v3(r), v3(r + 3), r[6], r[7]}; }

}  // namespace

// ===========================================================================
// C interface
// ===========================================================================
extern "C" {

void yo_rng_stream(uint64_t seed, uint64_t seq, int n, uint64_t* state_inc, float* out) {
  auto rng     = make_rng(seed, seq);
  state_inc[0] = rng.state, state_inc[1] = rng.inc;
  for (int i = 0; i < n; i++) out[i] = rand1f(rng);
}
void yo_pixel_seqs(int n, int* out) {
  auto rng = make_rng(1301081);
  for (int i = 0; i < n; i++) out[i] = rand1i_2p31(rng) / 2 + 1;
}
void yo_hair_brdf(int n, const float* mats, const float* v, const float* nrm,
    const float* tng, float* out) {
  for (int i = 0; i < n; i++) {
    auto b = eval_hair_brdf(mkhair(mats + 12 * i), v[i], v3(nrm + 3 * i), v3(tng + 3 * i));
    std::memcpy(out + 30 * i, &b, 120);
  }
}
void yo_hair_eval(int n, const float* brdf, const float* wo, const float* wi, float* out) {
  for (int i = 0; i < n; i++) {
    auto f = eval_hair_scattering(mkbrdf(brdf + 30 * i), v3(wo + 3 * i), v3(wi + 3 * i));
    out[3 * i] = f.x, out[3 * i + 1] = f.y, out[3 * i + 2] = f.z;
  }
}
void yo_hair_sample(int n, const float* brdf, const float* wo, const float* rn, float* out) {
  for (int i = 0; i < n; i++) {
    auto w = sample_hair_scattering(mkbrdf(brdf + 30 * i), v3(wo + 3 * i), rn[2 * i], rn[2 * i + 1]);
    out[3 * i] = w.x, out[3 * i + 1] = w.y, out[3 * i + 2] = w.z;
  }
}
void yo_hair_pdf(int n, const float* brdf, const float* wo, const float* wi, float* out) {
  for (int i = 0; i < n; i++)
    out[i] = sample_hair_scattering_pdf(mkbrdf(brdf + 30 * i), v3(wo + 3 * i), v3(wi + 3 * i));
}

// The four self-tests (ext.cpp:555-693): same seed, loop bounds (float
// accumulating loop counters included), sample counts and thresholds.
int yo_selftest(int which, float* worst) {
  auto rng  = make_rng(199382389514);
  auto ok   = true;
  auto dev  = 0.0f;  // largest deviation from the target statistic
  auto mkmat = [](float beta_m, float beta_n) {
    HairMaterial m{};
    m.sigma_a = {0, 0, 0}, m.color = {0, 0, 0};
    m.beta_m = beta_m, m.beta_n = beta_n, m.alpha = 0, m.eta = 1.55f;
    m.eumelanin = 0, m.pheomelanin = 0;
    return m;
  };
  if (which == 0 || which == 1) {
    auto wx = rand1f(rng), wy = rand1f(rng);
    auto wo = sample_sphere(wx, wy);
    for (auto beta_m = 0.1f; beta_m < 1.0f; beta_m += 0.2f) {
      for (auto beta_n = 0.1f; beta_n < 1.0f; beta_n += 0.2f) {
        auto sum   = V3{0, 0, 0};
        auto count = 300000;
        for (auto i = 0; i < count; i++) {
          auto h = rand1f(rng);
          if (which == 0 && h == 0) h += flt_eps;
          auto brdf = eval_hair_brdf(mkmat(beta_m, beta_n), h, {0, 0, 1}, {1, 0, 0});
          auto rx = rand1f(rng), ry = rand1f(rng);
          if (which == 0) {
            auto wi = sample_sphere(rx, ry);
            sum     = sum + eval_hair_scattering(brdf, wo, wi);
          } else {
            auto wi  = sample_hair_scattering(brdf, wo, rx, ry);
            auto f   = eval_hair_scattering(brdf, wo, wi);
            auto pdf = sample_hair_scattering_pdf(brdf, wo, wi);
            if (pdf > 0) sum = sum + f / pdf;
          }
        }
        auto avg = which == 0 ? luminance(sum) / (count * (1 / (4 * pif)))
                              : luminance(sum) / (count);
        auto lo = which == 0 ? 0.95f : 0.99f, hi = which == 0 ? 1.05f : 1.01f;
        if (!(avg >= lo && avg <= hi)) ok = false;
        dev = fmax_(dev, fabs_(avg - 1));
      }
    }
  } else if (which == 2) {
    for (auto beta_m = 0.1f; beta_m < 1.0f; beta_m += 0.2f) {
      for (auto beta_n = 0.4f; beta_n < 1.0f; beta_n += 0.2f) {
        for (auto i = 0; i < 10000; i++) {
          auto h    = rand1f(rng);
          auto brdf = eval_hair_brdf(mkmat(beta_m, beta_n), h, {0, 0, 1}, {1, 0, 0});
          auto wx = rand1f(rng), wy = rand1f(rng);
          auto wo = sample_sphere(wx, wy);
          auto rx = rand1f(rng), ry = rand1f(rng);
          auto wi  = sample_hair_scattering(brdf, wo, rx, ry);
          auto f   = eval_hair_scattering(brdf, wo, wi);
          auto pdf = sample_hair_scattering_pdf(brdf, wo, wi);
          if (pdf > 0) {
            auto r = luminance(f) / pdf;
            if (!(r >= 0.999f && r <= 1.001f)) ok = false;
            dev = fmax_(dev, fabs_(r - 1));
          }
        }
      }
    }
  } else if (which == 3) {
    for (auto beta_m = 0.2f; beta_m < 1.0f; beta_m += 0.2f)
      for (auto beta_n = 0.4f; beta_n < 1.0f; beta_n += 0.2f) {
        const auto count = 64 * 1024;
        auto wx = rand1f(rng), wy = rand1f(rng);
        auto wo = sample_sphere(wx, wy);
        auto li = [](V3 w) { return V3{w.z * w.z, w.z * w.z, w.z * w.z}; };
        auto f_importance = V3{0, 0, 0}, f_uniform = V3{0, 0, 0};
        for (auto i = 0; i < count; i++) {
          auto h    = rand1f(rng);
          auto brdf = eval_hair_brdf(mkmat(beta_m, beta_n), h, {0, 0, 1}, {1, 0, 0});
          auto ux = rand1f(rng), uy = rand1f(rng);
          auto wi  = sample_hair_scattering(brdf, wo, ux, uy);
          auto f   = eval_hair_scattering(brdf, wo, wi);
          auto pdf = sample_hair_scattering_pdf(brdf, wo, wi);
          if (pdf > 0) f_importance = f_importance + f * li(wi) / (count * pdf);
          wi        = sample_sphere(ux, uy);
          f_uniform = f_uniform + eval_hair_scattering(brdf, wo, wi) * li(wi) /
                                      (count * (1 / (4 * pif)));
        }
        auto err = fabs_(luminance(f_importance) - luminance(f_uniform)) /
                   luminance(f_uniform);
        if (err >= 0.05f) ok = false;
        dev = fmax_(dev, err);
      }
  } else {
    return 0;
  }
  if (worst) *worst = dev;
  return ok ? 1 : 0;
}

void yo_intersect_line(int n, const float* rays, const float* p0, const float* p1,
    const float* r0, const float* r1, int* hit, float* uv, float* dist) {
  for (int i = 0; i < n; i++) {
    float u[2] = {0, 0}, d = 0;
    hit[i] = intersect_line(mkray(rays + 8 * i), v3(p0 + 3 * i), v3(p1 + 3 * i), r0[i], r1[i], u, d);
    uv[2 * i] = u[0], uv[2 * i + 1] = u[1], dist[i] = d;
  }
}
void yo_intersect_triangle(int n, const float* rays, const float* p0, const float* p1,
    const float* p2, int* hit, float* uv, float* dist) {
  for (int i = 0; i < n; i++) {
    float u[2] = {0, 0}, d = 0;
    hit[i] = intersect_triangle(mkray(rays + 8 * i), v3(p0 + 3 * i), v3(p1 + 3 * i), v3(p2 + 3 * i), u, d);
    uv[2 * i] = u[0], uv[2 * i + 1] = u[1], dist[i] = d;
  }
}
void yo_intersect_bbox(int n, const float* rays, const float* bbox, int* hit) {
  for (int i = 0; i < n; i++) {
    auto r    = mkray(rays + 8 * i);
    auto dinv = V3{1 / r.d.x, 1 / r.d.y, 1 / r.d.z};
    hit[i]    = intersect_bbox(r, dinv, BBox{v3(bbox + 6 * i), v3(bbox + 6 * i + 3)});
  }
}

// One lobe of yocto_math.h:4427-4755 per call. params: 8 floats per item (ior,
// roughness, eta[3], etak[3]); rn: 3 per item (rnl, rn.x, rn.y); out: 7 per
// item (value*|cos| [3], pdf, sampled incoming [3]).
void yo_surface_lobe(int kind, int n, const float* params, const float* normal,
    const float* outgoing, const float* incoming, const float* rn, float* out) {
  for (int i = 0; i < n; i++) {
    auto  q   = params + 8 * i;
    auto  ior = q[0], rough = q[1];
    auto  eta = v3(q + 2), etak = v3(q + 5);
    auto  nn = v3(normal + 3 * i), wo = v3(outgoing + 3 * i), wi = v3(incoming + 3 * i);
    auto  rnl = rn[3 * i], rx = rn[3 * i + 1], ry = rn[3 * i + 2];
    V3    f{0, 0, 0}, w{0, 0, 0};
    float pdf = 0;
    switch (kind) {
      case YH_LOBE_DIFFUSE:
        f = eval_diffuse_reflection(nn, wo, wi), pdf = sample_diffuse_reflection_pdf(nn, wo, wi);
        w = sample_diffuse_reflection(nn, wo, rx, ry);
        break;
      case YH_LOBE_SPECULAR:
        f   = eval_microfacet_reflection(ior, rough, nn, wo, wi);
        pdf = sample_microfacet_reflection_pdf(rough, nn, wo, wi);
        w   = sample_microfacet_reflection(rough, nn, wo, rx, ry);
        break;
      case YH_LOBE_METAL:
        f   = eval_microfacet_reflection(eta, etak, rough, nn, wo, wi);
        pdf = sample_microfacet_reflection_pdf(rough, nn, wo, wi);
        w   = sample_microfacet_reflection(rough, nn, wo, rx, ry);
        break;
      case YH_LOBE_TRANSMISSION:
        f   = eval_microfacet_transmission(ior, rough, nn, wo, wi);
        pdf = sample_microfacet_transmission_pdf(rough, nn, wo, wi);
        w   = sample_microfacet_transmission(rough, nn, wo, rx, ry);
        break;
      case YH_LOBE_REFRACTION:
        f   = eval_microfacet_refraction(ior, rough, nn, wo, wi);
        pdf = sample_microfacet_refraction_pdf(ior, rough, nn, wo, wi);
        w   = sample_microfacet_refraction(ior, rough, nn, wo, rnl, rx, ry);
        break;
      case YH_LOBE_DELTA_SPECULAR:
        f = eval_delta_reflection(ior, nn, wo, wi), pdf = sample_delta_reflection_pdf(nn, wo, wi);
        w = sample_delta_reflection(nn, wo);
        break;
      case YH_LOBE_DELTA_METAL:
        f = eval_delta_reflection(eta, etak, nn, wo, wi), pdf = sample_delta_reflection_pdf(nn, wo, wi);
        w = sample_delta_reflection(nn, wo);
        break;
      case YH_LOBE_DELTA_TRANSMISSION:
        f = eval_delta_transmission(nn, wo, wi), pdf = sample_delta_transmission_pdf(nn, wo, wi);
        w = sample_delta_transmission(nn, wo);
        break;
      case YH_LOBE_DELTA_REFRACTION:
        f = eval_delta_refraction(ior, nn, wo, wi), pdf = sample_delta_refraction_pdf(ior, nn, wo, wi);
        w = sample_delta_refraction(ior, nn, wo, rnl);
        break;
      default: break;
    }
    auto o = out + 7 * i;
    o[0] = f.x, o[1] = f.y, o[2] = f.z, o[3] = pdf, o[4] = w.x, o[5] = w.y, o[6] = w.z;
  }
}

// pbrt "curve" -> five-vertex line strand (yocto_pbrt.h:1751-1797, number_sub = 4;
// interpolate_bezier / _derivative math.h:3346-3357, lerp :1803). P: 12 floats per
// curve (the first four control points); out: positions 15, normals (= tangents) 15,
// radius 5 (= the pbrt width, as the reference stores it), lines 8 ints per curve with
// vertex indices starting at 5 * curve + base_vertex.
void yo_curves_to_lines(int n, const float* P, const float* width0, const float* width1,
    int base_vertex, float* positions, float* normals, float* radius, int* lines) {
  const int number_sub = 4;
  for (int c = 0; c < n; c++) {
    auto p0 = v3(P + 12 * c), p1 = v3(P + 12 * c + 3), p2 = v3(P + 12 * c + 6), p3 = v3(P + 12 * c + 9);
    auto put = [](float* o, V3 a) { o[0] = a.x, o[1] = a.y, o[2] = a.z; };
    for (int i = 0; i <= number_sub; i++) {
      V3    pos, tan;
      float rad;
      if (i == 0) {
        pos = p0, tan = normalize(p1 - p0), rad = width0[c];
      } else if (i == number_sub) {
        pos = p3, tan = normalize(p3 - p2), rad = width1[c];
      } else {
        auto u = (float)i / number_sub;
        pos    = p0 * (1 - u) * (1 - u) * (1 - u) + p1 * 3 * u * (1 - u) * (1 - u) + p2 * 3 * u * u * (1 - u) +
              p3 * u * u * u;
        tan = normalize((p1 - p0) * 3 * (1 - u) * (1 - u) + (p2 - p1) * 6 * u * (1 - u) + (p3 - p2) * 3 * u * u);
        rad = width0[c] * (1 - u) + width1[c] * u;
      }
      put(positions + 15 * c + 3 * i, pos), put(normals + 15 * c + 3 * i, tan);
      radius[5 * c + i] = rad;
    }
    for (int i = 0; i < number_sub; i++)
      lines[8 * c + 2 * i] = base_vertex + 5 * c + i, lines[8 * c + 2 * i + 1] = base_vertex + 5 * c + i + 1;
  }
}

// fresnel_dielectric, fresnel_conductor, reflectivity_to_eta(params.eta) — out 7n
void yo_fresnel(int n, const float* params, const float* normal, const float* outgoing, float* out) {
  for (int i = 0; i < n; i++) {
    auto q  = params + 8 * i;
    auto nn = v3(normal + 3 * i), wo = v3(outgoing + 3 * i);
    auto fd = fresnel_dielectric(q[0], nn, wo);
    auto fc = fresnel_conductor(v3(q + 2), v3(q + 5), nn, wo);
    auto e  = reflectivity_to_eta(v3(q + 2));
    auto o  = out + 7 * i;
    o[0] = fd, o[1] = fc.x, o[2] = fc.y, o[3] = fc.z, o[4] = e.x, o[5] = e.y, o[6] = e.z;
  }
}

// eval_brdf + the lobe dispatch of pt.cpp:405-471,1069-1280 for non-hair
// materials. out: YH_SURFACE_BSDF_FLOATS per item (layout in yhair.h).
void yo_surface_bsdf(int n, const yh_material* materials, const float* normal,
    const float* outgoing, const float* incoming, const float* rn, float* out) {
  for (int i = 0; i < n; i++) {
    auto&    m = materials[i];
    Material mt;
    mt.emission = v3(m.emission), mt.color = v3(m.color), mt.thin = m.thin != 0;
    mt.specular = m.specular, mt.metallic = m.metallic, mt.roughness = m.roughness;
    mt.ior = m.ior, mt.transmission = m.transmission, mt.opacity = m.opacity;
    auto nn = v3(normal + 3 * i), wo = v3(outgoing + 3 * i), wi = v3(incoming + 3 * i);
    auto rnl = rn[3 * i], rx = rn[3 * i + 1], ry = rn[3 * i + 2];
    auto b  = surface_brdf(mt, nn, wo);
    auto o  = out + YH_SURFACE_BSDF_FLOATS * i;
    V3   lobes[5] = {b.diffuse, b.specular, b.metal, b.transmission, b.refraction};
    for (int k = 0; k < 5; k++) o[3 * k] = lobes[k].x, o[3 * k + 1] = lobes[k].y, o[3 * k + 2] = lobes[k].z;
    o[15] = b.roughness, o[16] = b.opacity;
    o[17] = b.diffuse_pdf, o[18] = b.specular_pdf, o[19] = b.metal_pdf, o[20] = b.transmission_pdf,
    o[21] = b.refraction_pdf;
    V3    f, w;
    float pdf;
    if (!is_delta(b)) {
      f = eval_brdfcos(b, nn, wo, wi), pdf = sample_brdfcos_pdf(b, nn, wo, wi);
      w = sample_brdfcos(b, nn, wo, rnl, rx, ry);
    } else {
      f = eval_delta(b, nn, wo, wi), pdf = sample_delta_pdf(b, nn, wo, wi);
      w = sample_delta(b, nn, wo, rnl);
    }
    o[22] = f.x, o[23] = f.y, o[24] = f.z, o[25] = pdf, o[26] = w.x, o[27] = w.y, o[28] = w.z;
  }
}

yo_scene* yo_scene_create(const yh_scene_desc* d) {
  auto sc = new yo_scene{};
  for (int i = 0; i < d->num_textures; i++) {
    auto&   t = d->textures[i];
    Texture tx;
    tx.w = t.width, tx.h = t.height;
    size_t n = (size_t)t.width * t.height;
    if (t.is_byte) {
      tx.colorb.assign((const unsigned char*)t.pixels, (const unsigned char*)t.pixels + 3 * n);
    } else {
      tx.colorf.resize(n);
      std::memcpy(tx.colorf.data(), t.pixels, sizeof(float) * 3 * n);
    }
    sc->textures.push_back(std::move(tx));
  }
  for (int i = 0; i < d->num_shapes; i++) {
    auto& s = d->shapes[i];
    Shape sh;
    sh.positions.resize(s.num_vertices);
    std::memcpy(sh.positions.data(), s.positions, sizeof(float) * 3 * s.num_vertices);
    if (s.normals) {
      sh.normals.resize(s.num_vertices);
      std::memcpy(sh.normals.data(), s.normals, sizeof(float) * 3 * s.num_vertices);
    }
    if (s.texcoords) sh.texcoords.assign(s.texcoords, s.texcoords + 2 * (size_t)s.num_vertices);
    if (s.radius) sh.radius.assign(s.radius, s.radius + s.num_vertices);
    else if (s.num_lines) sh.radius.assign(s.num_vertices, 0.001f);  // sceneio.cpp:390
    if (s.num_lines) sh.lines.assign(s.lines, s.lines + 2 * s.num_lines);
    else if (s.num_triangles) sh.triangles.assign(s.triangles, s.triangles + 3 * s.num_triangles);
    init_shape_bvh(sh);
    sc->shapes.push_back(std::move(sh));
  }
  for (int i = 0; i < d->num_materials; i++) {
    auto& m = d->materials[i];
    Material mt;
    mt.emission = v3(m.emission), mt.color = v3(m.color);
    mt.thin = m.thin != 0;
    mt.specular = m.specular, mt.metallic = m.metallic, mt.roughness = m.roughness;
    mt.ior = m.ior, mt.transmission = m.transmission, mt.opacity = m.opacity;
    mt.scattering = v3(m.scattering), mt.scanisotropy = m.scanisotropy, mt.trdepth = m.trdepth;
    mt.emission_tex = m.emission_tex - 1, mt.color_tex = m.color_tex - 1, mt.scattering_tex = m.scattering_tex - 1;
    mt.hair.sigma_a = v3(m.sigma_a);
    mt.hair.beta_m = m.beta_m, mt.hair.beta_n = m.beta_n;
    mt.hair.alpha = m.alpha, mt.hair.eta = m.eta;
    mt.hair.color = v3(m.color);  // pt.cpp:482
    mt.hair.eumelanin = m.eumelanin, mt.hair.pheomelanin = m.pheomelanin;
    sc->materials.push_back(mt);
  }
  for (int i = 0; i < d->num_objects; i++)
    sc->objects.push_back({mkframe(d->objects[i].frame), d->objects[i].shape, d->objects[i].material});
  for (int i = 0; i < d->num_environments; i++) {
    auto& e = d->environments[i];
    Environment env;
    env.frame = mkframe(e.frame), env.emission = v3(e.emission);
    if (e.texels) {
      env.w = e.tex_width, env.h = e.tex_height;
      env.texels.resize((size_t)env.w * env.h);
      std::memcpy(env.texels.data(), e.texels, sizeof(float) * 3 * env.texels.size());
    }
    sc->environments.push_back(std::move(env));
  }
  auto& c    = d->camera;
  sc->camera = {mkframe(c.frame), c.lens, {c.film[0], c.film[1]}, c.focus, c.aperture};
  // scene-level BVH over instances (pt.cpp:792-814)
  auto prims = std::vector<BvhPrim>{};
  for (auto oid = 0; oid < (int)sc->objects.size(); oid++) {
    auto& obj = sc->objects[oid];
    auto& sh  = sc->shapes[obj.shape];
    BvhPrim p;
    p.bbox      = sh.bvh.nodes.empty() ? BBox{} : transform_bbox(obj.frame, sh.bvh.nodes[0].bbox);
    p.center    = center(p.bbox);
    p.primitive = oid;
    prims.push_back(p);
  }
  build_bvh(sc->bvh, prims);
  init_lights(*sc);
  return sc;
}
void yo_scene_free(yo_scene* s) { delete s; }
int  yo_scene_num_lights(const yo_scene* s) { return (int)s->lights.size(); }

void yo_scene_intersect(const yo_scene* scene, int n, const float* rays, int* object,
    int* element, float* uv, float* dist) {
  for (int i = 0; i < n; i++) {
    int   o = -1, e = -1;
    float u[2] = {0, 0}, d = 0;
    auto  hit = intersect_scene_bvh(*scene, mkray(rays + 8 * i), o, e, u, d);
    object[i] = hit ? o : -1, element[i] = hit ? e : -1;
    uv[2 * i] = u[0], uv[2 * i + 1] = u[1], dist[i] = d;
  }
}

// per-ray traversal work of the reference algorithm: nodes visited and segment
// / triangle tests, for load-balance and kernel-efficiency analysis
void yo_scene_intersect_counted(const yo_scene* scene, int n, const float* rays, int* nodes, int* prims) {
  for (int i = 0; i < n; i++) {
    tls_counters = Counters{};
    int   o = -1, e = -1;
    float u[2] = {0, 0}, d = 0;
    intersect_scene_bvh(*scene, mkray(rays + 8 * i), o, e, u, d);
    nodes[i] = (int)tls_counters.nodes, prims[i] = (int)(tls_counters.seg + tls_counters.tri);
  }
}

int yo_scene_bvh(const yo_scene* scene, int shape, float* nodes, int* prims) {
  auto& t = shape < 0 ? scene->bvh : scene->shapes[shape].bvh;
  if (nodes) {
    for (size_t i = 0; i < t.nodes.size(); i++) {
      auto& n = t.nodes[i];
      float* o = nodes + 8 * i;
      o[0] = n.bbox.min.x, o[1] = n.bbox.min.y, o[2] = n.bbox.min.z;
      o[3] = n.bbox.max.x, o[4] = n.bbox.max.y, o[5] = n.bbox.max.z;
      int a = n.start, b = (int)n.num | ((int)n.internal << 16) | ((int)n.axis << 24);
      std::memcpy(o + 6, &a, 4), std::memcpy(o + 7, &b, 4);
    }
  }
  if (prims) std::memcpy(prims, t.primitives.data(), sizeof(int) * t.primitives.size());
  return (int)t.nodes.size();
}

int yo_render(const yo_scene* scene, const yh_trace_params* params, int samples,
    int nthreads, int* width, int* height, float* rgba, uint64_t* rng_out,
    yh_workcounts* counts) {
  int w, h;
  if (params->shader < 0 || params->shader >= YH_SHADER_COUNT) return 1;  // "sampler unknown" (pt.cpp:1669)
  image_size(scene->camera, params->resolution, w, h);
  *width = w, *height = h;
  if (!rgba) return 0;
  auto pixels = std::vector<Pixel>((size_t)w * h);
  auto master = make_rng(1301081);  // pt.cpp:1942-1945
  for (auto& p : pixels) p.rng = make_rng(params->seed, rand1i_2p31(master) / 2 + 1);
  if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
  Counters   total;
  std::mutex mtx;
  for (int s = 0; s < samples; s++) {
    // parallel_for over rows with an atomic row counter (pt.cpp:1954-1970)
    std::atomic<int>         next_idx(0);
    std::vector<std::thread> threads;
    auto work = [&]() {
      tls_counters = Counters{};
      while (true) {
        auto j = next_idx.fetch_add(1);
        if (j >= h) break;
        for (auto i = 0; i < w; i++) {
          auto r = trace_sample(*scene, pixels[(size_t)j * w + i], i, j, w, h,
              params->bounces, params->clamp, params->shader);
          auto o = rgba + 4 * ((size_t)j * w + i);
          o[0] = r.x, o[1] = r.y, o[2] = r.z, o[3] = r.w;
        }
      }
      std::lock_guard<std::mutex> lock(mtx);
      auto& c = tls_counters;
      total.rays += c.rays, total.nodes += c.nodes, total.seg += c.seg;
      total.tri += c.tri, total.hair += c.hair, total.surf += c.surf;
      total.envl += c.envl, total.envs += c.envs, total.samples += c.samples;
    };
    if (nthreads == 1) {
      work();
    } else {
      for (int t = 0; t < nthreads; t++) threads.emplace_back(work);
      for (auto& t : threads) t.join();
    }
  }
  if (rng_out)
    for (size_t i = 0; i < pixels.size(); i++)
      rng_out[2 * i] = pixels[i].rng.state, rng_out[2 * i + 1] = pixels[i].rng.inc;
  if (counts) {
    std::memset(counts, 0, sizeof(*counts));
    counts->samples = total.samples, counts->rays = total.rays;
    counts->nodes = total.nodes, counts->seg_tests = total.seg;
    counts->tri_tests = total.tri, counts->hair_shades = total.hair;
    counts->surf_shades = total.surf, counts->env_lookups = total.envl;
    counts->env_samples = total.envs;
  }
  return 0;
}

}  // extern "C"
