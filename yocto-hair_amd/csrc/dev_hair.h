// dev_hair.h — the hair BSDF on the device (gfx950).
//
// Restates libs/yocto_extension/yocto_extension.cpp:90-551 (the pbrt-v3 /
// Chiang et al. hair model as the reference implements it):
//   eval_hair_brdf  (127-177)   -> hair_setup (per-hit part) + yhd_material
//                                  (per-material part, computed at upload)
//   eval_hair_scattering (255-336), sample_hair_scattering (399-479),
//   sample_hair_scattering_pdf (481-551) -> hair_eval / hair_sample /
//   hair_pdf, plus the fused hair_eval_pdf the integrator uses: f and pdf
//   share Mp and Np (same expressions, so the fused result is bit-identical
//   to calling the two separately).
//
// Reference quirks kept on purpose:
//   * asin and sinh resolve to the C double overloads in the reference
//     (ext.cpp:111,206); gamma_o / gamma_t use the double asin here too and
//     the v > 0.1 branch of Mp divides in double by the host-computed
//     sinh(1/v) * 2 * v.
//   * compute_ap_pdf re-derives sin_theta_o as sqrt(1 - cos^2) (ext.cpp:372),
//     so the Ap used for the pdf is NOT the Ap used for f; both are computed.
//   * the i0 series divides by int64 products (ext.cpp:179-192); the ten
//     divisors are exact in float, so they are literal constants here.
#ifndef YH_DEV_HAIR_H_
#define YH_DEV_HAIR_H_
#include "dev_math.h"

namespace yhd {

constexpr int p_max = 3;  // ext.h:84

// Per-hit part of hair_brdf (ext.h:97-113).
struct hair_hit {
  float h, gamma_o;
  frame w2b;  // world_to_brdf: rows of the local frame, o = -0
};

// Arithmetic of the BSDF. Its values are pinned to the reference within 1e-4 relative (sampled
// directions 5e-5), not bit for bit: libm differs between glibc and the device in the last ulp
// anyway, so paths through hair leave the reference's after a bounce or two regardless. Inside
// that tolerance the hardware's 1-ulp reciprocal / square root / log2 / sin / cos and asinf replace
// the IEEE division sequences and the library calls (YH_HAIR_FAST=0: the exact forms). Kept exact:
// everything that enters Mp's exponent — the local frame, sin / cos theta of both directions, the
// tilt, a = cos cos / v, b = sin sin / v and log I0 — because for a narrow longitudinal lobe
// (beta_m 0.05: 1 / v = 680) the exponent cancels terms of that magnitude and one ulp of an input
// is 1e-4 of the result. Geometry — traversal, intersection, positions, camera rays, the surface
// lobes — stays exact throughout.
#ifndef YH_HAIR_FAST
#define YH_HAIR_FAST 1 /* 0: the exact arithmetic (csrc/exact.hip compiles the sample loop a second time with it: yh_trace_params::hair_exact) */
#endif
// (what YH_HAIR_FAST stands for; the per-function A/Bs of round 2 are closed: profiles/r02/bsdf_arithmetic_variants.txt)
#define YH_FAST_DIV YH_HAIR_FAST
#define YH_FAST_EXP 0
#define YH_FAST_LOG YH_HAIR_FAST
#define YH_FAST_TRIG YH_HAIR_FAST
#define YH_FAST_ASIN YH_HAIR_FAST
YH_DEV float h_rcp(float x) { return YH_FAST_DIV ? __builtin_amdgcn_rcpf(x) : 1 / x; }
YH_DEV float h_div(float a, float b) { return YH_FAST_DIV ? a * __builtin_amdgcn_rcpf(b) : a / b; }
// quotient that feeds an exponential: one Newton step on the reciprocal (0.5 ulp), so that the
// relative error does not grow with the magnitude of the exponent
YH_DEV float h_div_n(float a, float b) {
  if (!YH_FAST_DIV) return a / b;
  float r = __builtin_amdgcn_rcpf(b);
  r       = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
  return a * r;
}
YH_DEV float h_sqrt(float x) { return YH_FAST_DIV ? __builtin_amdgcn_sqrtf(x) : sqrtf(x); }
// exp stays the library's: v_exp_f32(x log2 e) loses |x| * 1e-7 of the result (the azimuthal
// logistic reaches exponents of -275), carrying the product's rounding error along costs the dense
// configs 5-10 %, and neither is faster on C1 (A/B on one box, profiles/r01/README.md)
YH_DEV float h_exp(float x) { return YH_FAST_EXP ? __builtin_amdgcn_exp2f(x * 1.44269504f) : expf(x); }
YH_DEV float h_log(float x) { return YH_FAST_LOG ? __builtin_amdgcn_logf(x) * 0.693147181f : logf(x); }
YH_DEV float h_sin(float x) { return YH_FAST_TRIG ? __builtin_amdgcn_sinf(x * 0.159154943f) : sinf(x); }  // |x| well inside 256 turns
YH_DEV float h_cos(float x) { return YH_FAST_TRIG ? __builtin_amdgcn_cosf(x * 0.159154943f) : cosf(x); }
YH_DEV float h_cos_turns(float t) { return YH_FAST_TRIG ? __builtin_amdgcn_cosf(t) : cosf(2 * pif * t); }  // cos(2 pi t)
// the reference's asin is the double overload (ext.cpp:111)
YH_DEV float h_asin(float x) { return YH_FAST_ASIN ? asinf(x) : (float)asin((double)x); }
YH_DEV f3    h_normalize(f3 a) {
  if (!YH_FAST_DIV) return normalize(a);
  float l2 = dot(a, a);
  return l2 != 0 ? a * __builtin_amdgcn_rsqf(l2) : a;
}
YH_DEV f3 h_div(f3 a, float b) { float r = h_rcp(b); return YH_FAST_DIV ? a * r : a / b; }
YH_DEV f3 h_transform_direction(const frame& a, f3 b) { return h_normalize(transform_vector(a, b)); }

// The chain that ends in the SAMPLED DIRECTION (gamma_t of the transmitted ray, the longitudinal / azimuthal sampling formulas, the
// rotation into world space) is ALWAYS in the exact forms: a last-place difference there sends the path to another fibre within a
// few bounces, one in the lobes' VALUES only scales the weight (round 4: path following on light hair 0.52 -> 0.21 of the seed floor
// for 1.5-2 % of C1's throughput, profiles/r04/direction_chain_ab.txt; the all-fast A/B half is gone with round 5).
YH_DEV float d_rcp(float x) { return 1 / x; }
YH_DEV float d_div(float a, float b) { return a / b; }
YH_DEV float d_sqrt(float x) { return sqrtf(x); }
YH_DEV float d_safe_sqrt(float x) { return d_sqrt(fmax_(0.0f, x)); }
YH_DEV float d_log(float x) { return logf(x); }
YH_DEV float d_sin(float x) { return sinf(x); }
YH_DEV float d_cos(float x) { return cosf(x); }
YH_DEV float d_cos_turns(float t) { return cosf(2 * pif * t); }
YH_DEV f3    d_transform_direction(const frame& a, f3 b) { return normalize(transform_vector(a, b)); }

YH_DEV float sqr(float v) { return v * v; }
YH_DEV float safe_asin(float x) { return h_asin(fclamp(x, -1.0f, 1.0f)); }
YH_DEV float safe_sqrt(float x) { return h_sqrt(fmax_(0.0f, x)); }
YH_DEV float exact_safe_sqrt(float x) { return sqrtf(fmax_(0.0f, x)); }

// ext.cpp:148-151,174 with math.h:2898-2903 (frame_fromzx) and the rigid
// inverse (transpose).
// QUAD: called by the four lanes of a quad with identical arguments (the integrator): the exact
// divisions of the normalisations are spread over the lanes (dev_math.h), same bits.
template <bool QUAD = false>
YH_DEV hair_hit hair_setup(float v, f3 normal, f3 tangent) {
  hair_hit hh;
  hh.h       = -1 + 2 * v;
  hh.gamma_o = safe_asin(hh.h);
  f3 z       = QUAD ? quad_normalize(normal) : normalize(normal);
  f3 x       = QUAD ? quad_orthonormalize(tangent, z) : orthonormalize(tangent, z);
  f3 y       = QUAD ? quad_normalize(cross(z, x)) : normalize(cross(z, x));
  hh.w2b     = transpose_rot(frame{x, y, z, mk3(0.0f)});
  return hh;
}

// i0 (ext.cpp:179-192): divisors 4^i (i!)^2 as exact floats.
YH_DEV float i0(float x) {
  const float den[10] = {1.0f, 4.0f, 64.0f, 2304.0f, 147456.0f, 14745600.0f,
      2123366400.0f, 416179814400.0f, 106542032486400.0f, 34519618525593600.0f};
  float val = 0, x2i = 1;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    // the series' terms are not amplified by 1 / v: reciprocal constants (YH_HAIR_FAST) instead of
    // seven IEEE divisions per evaluation
    val += YH_FAST_DIV ? x2i * (1.0f / den[i]) : x2i / den[i];
    x2i *= x * x;
  }
  return val;
}
YH_DEV float log_i0(float x) {  // ext.cpp:194-199
  if (x > 12)
    return x + 0.5f * (-logf(2 * pif) + logf(1 / x) + 1 / (8 * x));
  else
    return logf(i0(x));
}
// mp (ext.cpp:201-207) for lobe p of material m
YH_DEV float mp(const yhd_material& m, int p, float cos_theta_i, float cos_theta_o,
    float sin_theta_i, float sin_theta_o) {
  float v = m.v[p];
  // exact divisions: for small v the exponent below cancels terms of magnitude 1 / v (hundreds), so
  // one ulp of a or b is 1e-5 of the result
  float a = cos_theta_i * cos_theta_o / v;
  float b = sin_theta_i * sin_theta_o / v;
  if (v <= 0.1f) {
    return h_exp(log_i0(a) - b - m.inv_v[p] + 0.6931f + m.log_inv_2v[p]);
  } else {
    if (YH_HAIR_FAST) return h_div(h_exp(-b) * i0(a), (float)m.mp_den[p]);
    return (float)((double)(expf(-b) * i0(a)) / m.mp_den[p]);
  }
}
// fresnel_dielectric with dot(normal, outgoing) = cos (math.h:4215-4235); exact form (surface lobes)
YH_DEV float fresnel_dielectric_cos(float eta, float cosw_) {
  float cosw  = fabs_(cosw_);
  float sin2  = 1 - cosw * cosw;
  float eta2  = eta * eta;
  float cos2t = 1 - sin2 / eta2;
  if (cos2t < 0) return 1;
  float t0 = sqrtf(cos2t);
  float t1 = eta * t0;
  float t2 = eta * cosw;
  float rs = (cosw - t1) / (cosw + t1);
  float rp = (t0 - t2) / (t0 + t2);
  return (rs * rs + rp * rp) / 2;
}
// the same inside the hair BSDF (ap)
YH_DEV float h_fresnel_dielectric_cos(float eta, float cosw_) {
  float cosw  = fabs_(cosw_);
  float sin2  = 1 - cosw * cosw;
  float eta2  = eta * eta;
  float cos2t = 1 - h_div(sin2, eta2);
  if (cos2t < 0) return 1;
  float t0 = h_sqrt(cos2t);
  float t1 = eta * t0;
  float t2 = eta * cosw;
  float rs = h_div(cosw - t1, cosw + t1);
  float rp = h_div(t0 - t2, t0 + t2);
  return (rs * rs + rp * rp) / 2;
}
// ap (ext.cpp:209-230)
YH_DEV void ap(float cos_theta_o, float eta, float h, f3 T, f3 out[p_max + 1]) {
  float cos_gamma_o = safe_sqrt(1 - h * h);
  float cos_theta   = cos_theta_o * cos_gamma_o;
  float f = h_fresnel_dielectric_cos(eta, 0.0f * 0.0f + 0.0f * 0.0f + 1.0f * cos_theta);
  out[0]  = mk3(f);
  out[1]  = sqr(1 - f) * T;
  out[2]  = out[1] * T * f;
  f3 den  = mk3(1.f) - T * f;
  f3 num  = out[2] * f * T;
  out[3]  = f3{h_div(num.x, den.x), h_div(num.y, den.y), h_div(num.z, den.z)};
}
// T for a given (sin_theta_o, cos_theta_o) (ext.cpp:281-291 / 375-384)
// (etap and sin_gamma_t in the exact forms: gamma_t is part of the sampled azimuth. Also where only T is used —
// the lobe pdfs — so that the one-lane and the quad forms, which share this code differently, keep computing the same bits)
YH_DEV f3 transmittance(const yhd_material& m, float h, float sin_theta_o, float cos_theta_o,
    float& gamma_t) {
  float sin_theta_t = h_div(sin_theta_o, m.eta);
  float cos_theta_t = safe_sqrt(1 - sqr(sin_theta_t));
  float etap        = d_div(d_sqrt(m.eta * m.eta - sqr(sin_theta_o)), cos_theta_o);
  float sin_gamma_t = d_div(h, etap);
  float cos_gamma_t = safe_sqrt(1 - sqr(sin_gamma_t));
  gamma_t           = safe_asin(sin_gamma_t);
  float k           = h_div_n(2 * cos_gamma_t, cos_theta_t);
  f3    sa          = ld3(m.sigma_a);
  return f3{h_exp(-sa.x * k), h_exp(-sa.y * k), h_exp(-sa.z * k)};
}
YH_DEV float phi_fn(int p, float gamma_o, float gamma_t) {  // ext.cpp:232-234
  return 2 * p * gamma_t - 2 * gamma_o + p * pif;
}
// np (ext.cpp:236-253): trimmed logistic of the wrapped azimuth difference
YH_DEV float np(const yhd_material& m, float phi, int p, float gamma_o, float gamma_t) {
  float dphi = phi - phi_fn(p, gamma_o, gamma_t);
  while (dphi > pif) dphi -= 2 * pif;
  while (dphi < -pif) dphi += 2 * pif;
  float x = fabs_(dphi);
  float e = h_exp(-h_div_n(x, m.s));
  return h_div(h_div(e, m.s * sqr(1 + e)), m.tl_norm);
}
// scale tilt of lobe p (ext.cpp:299-322)
YH_DEV void tilt(const yhd_material& m, int p, float sin_theta_o, float cos_theta_o,
    float& sin_theta_op, float& cos_theta_op) {
  if (p == 0) {
    sin_theta_op = sin_theta_o * m.cos_2k_alpha[1] - cos_theta_o * m.sin_2k_alpha[1];
    cos_theta_op = cos_theta_o * m.cos_2k_alpha[1] + sin_theta_o * m.sin_2k_alpha[1];
  } else if (p == 1) {
    sin_theta_op = sin_theta_o * m.cos_2k_alpha[0] + cos_theta_o * m.sin_2k_alpha[0];
    cos_theta_op = cos_theta_o * m.cos_2k_alpha[0] - sin_theta_o * m.sin_2k_alpha[0];
  } else if (p == 2) {
    sin_theta_op = sin_theta_o * m.cos_2k_alpha[2] + cos_theta_o * m.sin_2k_alpha[2];
    cos_theta_op = cos_theta_o * m.cos_2k_alpha[2] - sin_theta_o * m.sin_2k_alpha[2];
  } else {
    sin_theta_op = sin_theta_o;
    cos_theta_op = cos_theta_o;
  }
}
// compute_ap_pdf (ext.cpp:365-397)
YH_DEV void compute_ap_pdf(const yhd_material& m, float h, float cos_theta_o,
    float ap_pdf[p_max + 1]) {
  float sin_theta_o = safe_sqrt(1 - cos_theta_o * cos_theta_o);
  float gamma_t;
  f3    T = transmittance(m, h, sin_theta_o, cos_theta_o, gamma_t);
  f3    apv[p_max + 1];
  ap(cos_theta_o, m.eta, h, T, apv);
  float sum_y = 0.0f;
#pragma unroll
  for (int i = 0; i <= p_max; i++) sum_y += luminance(apv[i]);
#pragma unroll
  for (int i = 0; i <= p_max; i++) ap_pdf[i] = h_div(luminance(apv[i]), sum_y);
}

// Fused eval_hair_scattering + sample_hair_scattering_pdf.
template <bool WANT_F, bool WANT_PDF>
YH_DEV void hair_eval_pdf(const yhd_material& m, const hair_hit& hh, f3 outgoing_, f3 incoming_,
    f3& f, float& pdf) {
  f3    outgoing    = transform_direction(hh.w2b, outgoing_);
  f3    incoming    = transform_direction(hh.w2b, incoming_);
  float sin_theta_o = outgoing.x;
  float cos_theta_o = exact_safe_sqrt(1 - sqr(sin_theta_o));
  float phi_o       = atan2f(outgoing.z, outgoing.y);
  float sin_theta_i = incoming.x;
  float cos_theta_i = exact_safe_sqrt(1 - sqr(sin_theta_i));
  float phi_i       = atan2f(incoming.z, incoming.y);
  float gamma_t;
  f3    T   = transmittance(m, hh.h, sin_theta_o, cos_theta_o, gamma_t);
  float phi = phi_i - phi_o;
  f3    apv[p_max + 1];
  float ap_pdf[p_max + 1];
  if (WANT_F) ap(cos_theta_o, m.eta, hh.h, T, apv);
  if (WANT_PDF) compute_ap_pdf(m, hh.h, cos_theta_o, ap_pdf);
  f   = mk3(0.0f);
  pdf = 0.0f;
#pragma unroll
  for (int p = 0; p < p_max; p++) {
    float sin_theta_op, cos_theta_op;
    tilt(m, p, sin_theta_o, cos_theta_o, sin_theta_op, cos_theta_op);
    cos_theta_op = fabs_(cos_theta_op);
    float mpv    = mp(m, p, cos_theta_i, cos_theta_op, sin_theta_i, sin_theta_op);
    float npv    = np(m, phi, p, hh.gamma_o, gamma_t);
    if (WANT_F) f = f + mpv * apv[p] * npv;
    if (WANT_PDF) pdf += mpv * ap_pdf[p] * npv;
  }
  float mpl = mp(m, p_max, cos_theta_i, cos_theta_o, sin_theta_i, sin_theta_o);
  if (WANT_F) f = f + h_div(mpl * apv[p_max], 2 * pif);
  if (WANT_PDF) pdf += mpl * ap_pdf[p_max] * (1 / (2 * pif));
}

// Everything of eval / sample / pdf that depends on the OUTGOING direction only
// (ext.cpp:267-273,281-295,365-397,410-421): computed once per shaded hit and
// shared by sample_hair_scattering and the fused eval + pdf — the reference
// recomputes the same expressions in each of the three functions, so sharing
// them changes no bit.
// (Members, not arrays: a per-lane `p` indexing an array member keeps the whole struct in scratch —
// 80 B written and re-read per shaded hit, 3.6 GB of HBM writes per C1 launch before this.)
struct hair_out {
  float sin_theta_o, cos_theta_o, phi_o, gamma_t;
  f3    ap0, ap1, ap2, ap3;      // Ap for f: T from sin_theta_o = outgoing.x (ext.cpp:281-295)
  float pdf0, pdf1, pdf2, pdf3;  // lobe pdfs: T from sin_theta_o = sqrt(1 - cos^2) (ext.cpp:372)
};
template <bool QUAD = false>
YH_DEV hair_out hair_prepare(const yhd_material& m, const hair_hit& hh, f3 outgoing_) {
  hair_out o;
  f3 outgoing   = QUAD ? quad_normalize(transform_vector(hh.w2b, outgoing_)) : transform_direction(hh.w2b, outgoing_);
  o.sin_theta_o = outgoing.x;
  o.cos_theta_o = exact_safe_sqrt(1 - sqr(o.sin_theta_o));
  o.phi_o       = atan2f(outgoing.z, outgoing.y);
  if (QUAD) {
    // The reference evaluates the transmittance and Ap twice, with sin_theta_o = outgoing.x for f
    // (ext.cpp:281-295) and with sqrt(1 - cos^2) for the lobe pdfs (ext.cpp:372-397): lanes 0-1 of
    // the quad take the first, lanes 2-3 the second, in one pass over the same code.
    const bool for_pdf = (__lane_id() & 2u) != 0;
    float      st      = for_pdf ? safe_sqrt(1 - o.cos_theta_o * o.cos_theta_o) : o.sin_theta_o;
    float      gt;
    f3         T = transmittance(m, hh.h, st, o.cos_theta_o, gt);
    f3         a[p_max + 1];
    ap(o.cos_theta_o, m.eta, hh.h, T, a);
    float l0 = luminance(a[0]), l1 = luminance(a[1]), l2 = luminance(a[2]), l3 = luminance(a[3]);
    float sum_y = 0.0f + l0 + l1 + l2 + l3;
    o.gamma_t = quad_bcast_f<0>(gt);
    o.ap0 = f3{quad_bcast_f<0>(a[0].x), quad_bcast_f<0>(a[0].y), quad_bcast_f<0>(a[0].z)};
    o.ap1 = f3{quad_bcast_f<0>(a[1].x), quad_bcast_f<0>(a[1].y), quad_bcast_f<0>(a[1].z)};
    o.ap2 = f3{quad_bcast_f<0>(a[2].x), quad_bcast_f<0>(a[2].y), quad_bcast_f<0>(a[2].z)};
    o.ap3 = f3{quad_bcast_f<0>(a[3].x), quad_bcast_f<0>(a[3].y), quad_bcast_f<0>(a[3].z)};
    o.pdf0 = quad_bcast_f<2>(h_div(l0, sum_y)), o.pdf1 = quad_bcast_f<2>(h_div(l1, sum_y));
    o.pdf2 = quad_bcast_f<2>(h_div(l2, sum_y)), o.pdf3 = quad_bcast_f<2>(h_div(l3, sum_y));
    return o;
  }
  f3 T          = transmittance(m, hh.h, o.sin_theta_o, o.cos_theta_o, o.gamma_t);
  f3    apv[p_max + 1];
  float ap_pdf[p_max + 1];
  ap(o.cos_theta_o, m.eta, hh.h, T, apv);
  compute_ap_pdf(m, hh.h, o.cos_theta_o, ap_pdf);
  o.ap0 = apv[0], o.ap1 = apv[1], o.ap2 = apv[2], o.ap3 = apv[3];
  o.pdf0 = ap_pdf[0], o.pdf1 = ap_pdf[1], o.pdf2 = ap_pdf[2], o.pdf3 = ap_pdf[3];
  return o;
}

// The integrator's fused eval + pdf with the four lobes p = 0..3 spread over the
// four lanes of a quad (dev_math.h): lane p evaluates Mp and Np of lobe p (the
// transcendental-heavy part: exp, log, the I0 series) and the lobe terms are
// summed in the reference's order p = 0, 1, 2, 3 (ext.cpp:297-332, 516-549) in
// every lane. Bit-identical to hair_eval_pdf<true, true>.
YH_DEV void hair_eval_pdf_quad(const yhd_material& m, const hair_hit& hh, const hair_out& ho, f3 incoming_,
    f3& f, float& pdf) {
  const int p       = (int)(__lane_id() & 3u);
  f3    incoming    = quad_normalize(transform_vector(hh.w2b, incoming_));  // quad-uniform up to here
  float sin_theta_o = ho.sin_theta_o, cos_theta_o = ho.cos_theta_o;
  float sin_theta_i = incoming.x;
  float cos_theta_i = exact_safe_sqrt(1 - sqr(sin_theta_i));
  float phi_i       = atan2f(incoming.z, incoming.y);
  float phi         = phi_i - ho.phi_o;
  // this lane's lobe
  float sin_theta_op, cos_theta_op;
  tilt(m, p, sin_theta_o, cos_theta_o, sin_theta_op, cos_theta_op);
  if (p < p_max) cos_theta_op = fabs_(cos_theta_op);
  float mpv  = mp(m, p, cos_theta_i, cos_theta_op, sin_theta_i, sin_theta_op);
  float npv  = p < p_max ? np(m, phi, p, hh.gamma_o, ho.gamma_t) : 0.0f;
  // values first, then selects: selecting between member loads becomes a load from a selected
  // address, which pins `ho` in scratch
  const f3    a0 = ho.ap0, a1 = ho.ap1, a2 = ho.ap2, a3 = ho.ap3;
  const float q0 = ho.pdf0, q1 = ho.pdf1, q2 = ho.pdf2, q3 = ho.pdf3;
  const bool  is0 = p == 0, is1 = p == 1, is2 = p == 2;
  f3    apq  = f3{is0 ? a0.x : is1 ? a1.x : is2 ? a2.x : a3.x, is0 ? a0.y : is1 ? a1.y : is2 ? a2.y : a3.y,
      is0 ? a0.z : is1 ? a1.z : is2 ? a2.z : a3.z};
  float appq = is0 ? q0 : is1 ? q1 : is2 ? q2 : q3;
  f3    tf   = p < p_max ? mpv * apq * npv : h_div(mpv * apq, 2 * pif);
  float tp   = p < p_max ? mpv * appq * npv : mpv * appq * (1 / (2 * pif));
  f   = mk3(0.0f);
  pdf = 0.0f;
  f   = f + f3{quad_bcast_f<0>(tf.x), quad_bcast_f<0>(tf.y), quad_bcast_f<0>(tf.z)};
  pdf += quad_bcast_f<0>(tp);
  f   = f + f3{quad_bcast_f<1>(tf.x), quad_bcast_f<1>(tf.y), quad_bcast_f<1>(tf.z)};
  pdf += quad_bcast_f<1>(tp);
  f   = f + f3{quad_bcast_f<2>(tf.x), quad_bcast_f<2>(tf.y), quad_bcast_f<2>(tf.z)};
  pdf += quad_bcast_f<2>(tp);
  f   = f + f3{quad_bcast_f<3>(tf.x), quad_bcast_f<3>(tf.y), quad_bcast_f<3>(tf.z)};
  pdf += quad_bcast_f<3>(tp);
}
// The same with ONE lane per path (YH_LANE, csrc/stream.hip): the four lobes one after the other, each
// term the expression of the quad form above and summed in the same order, so the same bits.
YH_DEV void hair_eval_pdf_lane(const yhd_material& m, const hair_hit& hh, const hair_out& ho, f3 incoming_,
    f3& f, float& pdf) {
  f3    incoming    = normalize(transform_vector(hh.w2b, incoming_));
  float sin_theta_o = ho.sin_theta_o, cos_theta_o = ho.cos_theta_o;
  float sin_theta_i = incoming.x;
  float cos_theta_i = exact_safe_sqrt(1 - sqr(sin_theta_i));
  float phi_i       = atan2f(incoming.z, incoming.y);
  float phi         = phi_i - ho.phi_o;
  f   = mk3(0.0f);
  pdf = 0.0f;
  const f3    apv[p_max]  = {ho.ap0, ho.ap1, ho.ap2};
  const float appv[p_max] = {ho.pdf0, ho.pdf1, ho.pdf2};
#pragma unroll
  for (int p = 0; p < p_max; p++) {
    float sin_theta_op, cos_theta_op;
    tilt(m, p, sin_theta_o, cos_theta_o, sin_theta_op, cos_theta_op);
    cos_theta_op = fabs_(cos_theta_op);
    float mpv    = mp(m, p, cos_theta_i, cos_theta_op, sin_theta_i, sin_theta_op);
    float npv    = np(m, phi, p, hh.gamma_o, ho.gamma_t);
    f            = f + mpv * apv[p] * npv;
    pdf += mpv * appv[p] * npv;
  }
  float mpl = mp(m, p_max, cos_theta_i, cos_theta_o, sin_theta_i, sin_theta_o);
  f         = f + h_div(mpl * ho.ap3, 2 * pif);
  pdf += mpl * ho.pdf3 * (1 / (2 * pif));
}
YH_DEV void hair_eval_pdf_quad(const yhd_material& m, const hair_hit& hh, f3 outgoing_, f3 incoming_,
    f3& f, float& pdf) {
  hair_out ho = hair_prepare<true>(m, hh, outgoing_);
  hair_eval_pdf_quad(m, hh, ho, incoming_, f, pdf);
}

// ext.cpp:339-357
YH_DEV uint32_t compact1by1(uint32_t x) {
  x &= 0x55555555;
  x = (x ^ (x >> 1)) & 0x33333333;
  x = (x ^ (x >> 2)) & 0x0f0f0f0f;
  x = (x ^ (x >> 4)) & 0x00ff00ff;
  x = (x ^ (x >> 8)) & 0x0000ffff;
  return x;
}
YH_DEV void demux_float(float f, float& a, float& b) {
  uint64_t v = (uint64_t)(f * 4294967296.0f);
  a          = (float)compact1by1((uint32_t)v) / 65536.0f;
  b          = (float)compact1by1((uint32_t)(v >> 1)) / 65536.0f;
}

// sample_hair_scattering (ext.cpp:399-479) given the outgoing-only terms
YH_DEV f3 hair_sample(const yhd_material& m, const hair_hit& hh, const hair_out& ho, float rnx, float rny) {
  float sin_theta_o = ho.sin_theta_o, cos_theta_o = ho.cos_theta_o, phi_o = ho.phi_o;
  float u00, u01, u10, u11;
  demux_float(rnx, u00, u01);
  demux_float(rny, u10, u11);
  // the reference's loop `for (p = 0; p < p_max; p++) { if (u[0][0] < ap_pdf[p]) break; u[0][0] -= ap_pdf[p]; }`
  int   p  = 0;
  float u1 = u00 - ho.pdf0, u2 = u1 - ho.pdf1;
  if (!(u00 < ho.pdf0)) p = !(u1 < ho.pdf1) ? (!(u2 < ho.pdf2) ? 3 : 2) : 1;
  float sin_theta_op, cos_theta_op;
  tilt(m, p, sin_theta_o, cos_theta_o, sin_theta_op, cos_theta_op);
  u10 = fmax_(u10, 1e-5f);
  float vp = 0, em2v = 0;
  // select without dynamic indexing into the struct
  vp   = p == 0 ? m.v[0] : p == 1 ? m.v[1] : m.v[2];  // v[3] == v[2]
  em2v = p == 0 ? m.exp_m2_inv_v[0] : p == 1 ? m.exp_m2_inv_v[1] : m.exp_m2_inv_v[2];
  float cos_theta   = 1 + vp * d_log(u10 + (1 - u10) * em2v);
  float sin_theta   = d_safe_sqrt(1 - sqr(cos_theta));
  float cos_phi     = d_cos_turns(u11);
  float sin_theta_i = -cos_theta * sin_theta_op + sin_theta * cos_phi * cos_theta_op;
  float cos_theta_i = d_safe_sqrt(1 - sqr(sin_theta_i));
  // gamma_t (ext.cpp:463-465) is the same expression as in eval: ho.gamma_t
  float dphi;
  if (p < p_max) {
    // sample_trimmed_logistic (ext.cpp:359-363)
    float x = -m.s * d_log(d_rcp(u01 * m.tl_norm + m.tl_cdf_a) - 1);
    dphi    = phi_fn(p, hh.gamma_o, ho.gamma_t) + fclamp(x, -pif, pif);
  } else {
    dphi = 2 * pif * u01;
  }
  float phi_i    = phi_o + dphi;
  f3    incoming = f3{sin_theta_i, cos_theta_i * d_cos(phi_i), cos_theta_i * d_sin(phi_i)};
  return d_transform_direction(transpose_rot(hh.w2b), incoming);
}
YH_DEV f3 hair_sample(const yhd_material& m, const hair_hit& hh, f3 outgoing_, float rnx, float rny) {
  hair_out ho = hair_prepare(m, hh, outgoing_);
  return hair_sample(m, hh, ho, rnx, rny);
}

// Fills the per-material constants from (beta-derived) v[], s on the DEVICE;
// used by the unit-level batch kernels whose hair_brdf arrives as 30 floats.
// The integrator uses the host-computed table instead.
YH_DEV void derive_material(yhd_material& m) {
#pragma unroll
  for (int p = 0; p <= p_max; p++) {
    m.inv_v[p]        = 1 / m.v[p];
    m.log_inv_2v[p]   = logf(1 / (2 * m.v[p]));
    m.exp_m2_inv_v[p] = expf(-2 / m.v[p]);
    m.mp_den[p]       = sinh((double)(1 / m.v[p])) * 2 * m.v[p];
  }
  float cb   = 1 / (1 + expf(-pif / m.s));
  float ca   = 1 / (1 + expf(-(-pif) / m.s));
  m.tl_cdf_a = ca;
  m.tl_norm  = cb - ca;
}

}  // namespace yhd
#endif
