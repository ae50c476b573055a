// dev_surface.h — the non-hair surface lobes.
//
// Restates yocto_math.h:4215-4273 (Fresnel terms), 4307-4375 (GGX microfacet
// distribution, shadowing, sampling), 4427-4755 (diffuse, microfacet
// reflection / transmission / refraction and the delta lobes) and the lobe
// mixture of pt.cpp:405-471 (eval_brdf) and 1069-1280 (eval / sample / pdf
// dispatch, rough and delta). Arithmetic is written in the reference's order,
// so everything built from + - * / sqrt (Fresnel terms, lobe weights, lobe
// pdfs, D, G) is bit-identical to the CPU; only atan / sin / cos in
// sample_microfacet can differ in the last ulp.
//
// The homogeneous medium inside a closed transmissive object (SURVEY.md 8(f)
// rank 2) follows at the end: pt.cpp:498-533,1360-1377 and math.h:4758-4822.
// Textures are not represented.
#ifndef YH_DEV_SURFACE_H_
#define YH_DEV_SURFACE_H_
#include "dev_hair.h"
#include "dev_math.h"

namespace yhd {

YH_DEV f3 reflect(f3 w, f3 n) { return -w + 2 * dot(n, w) * n; }  // math.h:2059
YH_DEV f3 refract(f3 w, f3 n, float inv_eta) {                    // math.h:2062
  float cosine = dot(n, w);
  float k      = 1 + inv_eta * inv_eta * (cosine * cosine - 1);
  if (k < 0) return mk3(0.0f);
  return -w * inv_eta + (inv_eta * cosine - sqrtf(k)) * n;
}

// fresnel_dielectric (math.h:4212-4235); the cosine form lives in dev_hair.h
YH_DEV float fresnel_dielectric(float eta, f3 normal, f3 outgoing) {
  return fresnel_dielectric_cos(eta, dot(normal, outgoing));
}
// fresnel_conductor (math.h:4238-4262), one channel
YH_DEV float conductor_channel(float eta, float etak, float cosw, float cos2, float sin2) {
  float eta2 = eta * eta, etak2 = etak * etak;
  float t0       = eta2 - etak2 - sin2;
  float a2plusb2 = sqrtf(t0 * t0 + 4 * eta2 * etak2);
  float t1       = a2plusb2 + cos2;
  float a        = sqrtf((a2plusb2 + t0) / 2);
  float t2       = 2 * a * cosw;
  float rs       = (t1 - t2) / (t1 + t2);
  float t3       = cos2 * a2plusb2 + sin2 * sin2;
  float t4       = t2 * sin2;
  float rp       = rs * (t3 - t4) / (t3 + t4);
  return (rp + rs) / 2;
}
YH_DEV f3 fresnel_conductor(f3 eta, f3 etak, f3 normal, f3 outgoing) {
  float cosw = dot(normal, outgoing);
  if (cosw <= 0) return mk3(0.0f);
  cosw       = fclamp(cosw, -1.0f, 1.0f);
  float cos2 = cosw * cosw;
  float sin2 = fclamp(1 - cos2, 0.0f, 1.0f);
  return {conductor_channel(eta.x, etak.x, cosw, cos2, sin2), conductor_channel(eta.y, etak.y, cosw, cos2, sin2),
      conductor_channel(eta.z, etak.z, cosw, cos2, sin2)};
}

// GGX pieces (math.h:4307-4375)
YH_DEV float microfacet_distribution(float roughness, f3 normal, f3 halfway) {
  float cosine = dot(normal, halfway);
  if (cosine <= 0) return 0;
  float roughness2 = roughness * roughness;
  float cosine2    = cosine * cosine;
  return roughness2 / (pif * (cosine2 * roughness2 + 1 - cosine2) * (cosine2 * roughness2 + 1 - cosine2));
}
YH_DEV float microfacet_shadowing1(float roughness, f3 normal, f3 halfway, f3 direction) {
  float cosine  = dot(normal, direction);
  float cosineh = dot(halfway, direction);
  if (cosine * cosineh <= 0) return 0;
  float roughness2 = roughness * roughness;
  float cosine2    = cosine * cosine;
  return 2 * fabs_(cosine) / (fabs_(cosine) + sqrtf(cosine2 - roughness2 * cosine2 + roughness2));
}
YH_DEV float microfacet_shadowing(float roughness, f3 normal, f3 halfway, f3 outgoing, f3 incoming) {
  return microfacet_shadowing1(roughness, normal, halfway, outgoing) *
         microfacet_shadowing1(roughness, normal, halfway, incoming);
}
// basis_fromz + transform_direction (math.h:2743-2752): local (z = normal) -> world
YH_DEV f3 from_local_z(f3 normal, f3 local) {
  f3    zz   = normalize(normal);
  float sign = copysignf(1.0f, zz.z);
  float a    = -1.0f / (sign + zz.z);
  float b    = zz.x * zz.y * a;
  f3    x    = {1.0f + sign * zz.x * zz.x * a, sign * b, -sign * zz.x};
  f3    y    = {b, sign + zz.y * zz.y * a, -zz.y};
  return normalize(x * local.x + y * local.y + zz * local.z);
}
YH_DEV f3 sample_microfacet(float roughness, f3 normal, float rx, float ry) {
  float phi   = 2 * pif * rx;
  float theta = atanf(roughness * sqrtf(ry / (1 - ry)));
  float st = sinf(theta), ct = cosf(theta);
  return from_local_z(normal, f3{cosf(phi) * st, sinf(phi) * st, ct});
}
YH_DEV float sample_microfacet_pdf(float roughness, f3 normal, f3 halfway) {
  float cosine = dot(normal, halfway);
  if (cosine < 0) return 0;
  return microfacet_distribution(roughness, normal, halfway) * cosine;
}
// sample_hemisphere_cos(normal, ruv) (math.h:4856-4878)
YH_DEV f3 sample_hemisphere_cos(f3 normal, float rx, float ry) {
  float z   = sqrtf(ry);
  float r   = sqrtf(1 - z * z);
  float phi = 2 * pif * rx;
  return from_local_z(normal, f3{r * cosf(phi), r * sinf(phi), z});
}
YH_DEV bool both_above(f3 normal, f3 outgoing, f3 incoming) {
  return !(dot(normal, incoming) <= 0 || dot(normal, outgoing) <= 0);
}

// ---- rough lobes: value * |cos|, sample, pdf (math.h:4427-4643) --------------
YH_DEV f3 eval_diffuse_reflection(f3 normal, f3 outgoing, f3 incoming) {
  if (!both_above(normal, outgoing, incoming)) return mk3(0.0f);
  return mk3(1.0f) / pif * dot(normal, incoming);
}
YH_DEV float sample_diffuse_reflection_pdf(f3 normal, f3 outgoing, f3 incoming) {
  if (!both_above(normal, outgoing, incoming)) return 0;
  float cosw = dot(normal, incoming);
  return (cosw <= 0) ? 0 : cosw / pif;
}
YH_DEV f3 sample_diffuse_reflection(f3 normal, f3 outgoing, float rx, float ry) {
  if (dot(normal, outgoing) <= 0) return mk3(0.0f);
  return sample_hemisphere_cos(normal, rx, ry);
}
// F as a vector: {f,f,f} for the dielectric lobe, the conductor term for metal
YH_DEV f3 microfacet_reflection_with(f3 F, float roughness, f3 normal, f3 halfway, f3 outgoing, f3 incoming) {
  float D = microfacet_distribution(roughness, normal, halfway);
  float G = microfacet_shadowing(roughness, normal, halfway, outgoing, incoming);
  return F * D * G / (4 * dot(normal, outgoing) * dot(normal, incoming)) * dot(normal, incoming);
}
YH_DEV f3 eval_microfacet_reflection(float ior, float roughness, f3 normal, f3 outgoing, f3 incoming) {
  if (!both_above(normal, outgoing, incoming)) return mk3(0.0f);
  f3    halfway = normalize(incoming + outgoing);
  float f       = fresnel_dielectric(ior, halfway, incoming);
  return microfacet_reflection_with(mk3(1.0f) * f, roughness, normal, halfway, outgoing, incoming);
}
YH_DEV f3 eval_microfacet_reflection(f3 eta, f3 etak, float roughness, f3 normal, f3 outgoing, f3 incoming) {
  if (!both_above(normal, outgoing, incoming)) return mk3(0.0f);
  f3 halfway = normalize(incoming + outgoing);
  return microfacet_reflection_with(fresnel_conductor(eta, etak, halfway, incoming), roughness, normal, halfway,
      outgoing, incoming);
}
YH_DEV float sample_microfacet_reflection_pdf(float roughness, f3 normal, f3 outgoing, f3 incoming) {
  if (!both_above(normal, outgoing, incoming)) return 0;
  f3 halfway = normalize(outgoing + incoming);
  return sample_microfacet_pdf(roughness, normal, halfway) / (4 * fabs_(dot(outgoing, halfway)));
}
YH_DEV f3 sample_microfacet_reflection(float roughness, f3 normal, f3 outgoing, float rx, float ry) {
  if (dot(normal, outgoing) <= 0) return mk3(0.0f);
  return reflect(outgoing, sample_microfacet(roughness, normal, rx, ry));
}
YH_DEV f3 eval_microfacet_transmission(float roughness, f3 normal, f3 outgoing, f3 incoming) {
  if (dot(normal, incoming) >= 0 || dot(normal, outgoing) <= 0) return mk3(0.0f);
  f3    reflected = reflect(-incoming, normal);
  f3    halfway   = normalize(reflected + outgoing);
  float D         = microfacet_distribution(roughness, normal, halfway);
  float G         = microfacet_shadowing(roughness, normal, halfway, outgoing, reflected);
  return mk3(1.0f) * D * G / (4 * dot(normal, outgoing) * dot(normal, reflected)) * (dot(normal, reflected));
}
YH_DEV float sample_microfacet_transmission_pdf(float roughness, f3 normal, f3 outgoing, f3 incoming) {
  if (dot(normal, incoming) >= 0 || dot(normal, outgoing) <= 0) return 0;
  f3    reflected = reflect(-incoming, normal);
  f3    halfway   = normalize(reflected + outgoing);
  float d         = sample_microfacet_pdf(roughness, normal, halfway);
  return d / (4 * fabs_(dot(outgoing, halfway)));
}
YH_DEV f3 sample_microfacet_transmission(float roughness, f3 normal, f3 outgoing, float rx, float ry) {
  if (dot(normal, outgoing) <= 0) return mk3(0.0f);
  f3 reflected = reflect(outgoing, sample_microfacet(roughness, normal, rx, ry));
  return -reflect(reflected, normal);
}
struct refraction_side {  // shared head of math.h:4486-4489 and friends
  bool  entering;
  f3    up_normal;
  float rel_ior;
};
YH_DEV refraction_side refraction_setup(float ior, f3 normal, f3 outgoing) {
  refraction_side s;
  s.entering  = dot(normal, outgoing) >= 0;
  s.up_normal = s.entering ? normal : -normal;
  s.rel_ior   = s.entering ? ior : (1 / ior);
  return s;
}
YH_DEV f3 eval_microfacet_refraction(float ior, float roughness, f3 normal, f3 outgoing, f3 incoming) {
  refraction_side s = refraction_setup(ior, normal, outgoing);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    f3    halfway = normalize(incoming + outgoing);
    float F       = fresnel_dielectric(s.rel_ior, halfway, outgoing);
    float D       = microfacet_distribution(roughness, s.up_normal, halfway);
    float G       = microfacet_shadowing(roughness, s.up_normal, halfway, outgoing, incoming);
    return mk3(1.0f) * F * D * G / fabs_(4 * dot(normal, outgoing) * dot(normal, incoming)) *
           fabs_(dot(normal, incoming));
  } else {
    f3    halfway = -normalize(s.rel_ior * incoming + outgoing) * (s.entering ? 1.0f : -1.0f);
    float F       = fresnel_dielectric(s.rel_ior, halfway, outgoing);
    float D       = microfacet_distribution(roughness, s.up_normal, halfway);
    float G       = microfacet_shadowing(roughness, s.up_normal, halfway, outgoing, incoming);
    float den     = s.rel_ior * dot(halfway, incoming) + dot(halfway, outgoing);
    return mk3(1.0f) *
           fabs_((dot(outgoing, halfway) * dot(incoming, halfway)) / (dot(outgoing, normal) * dot(incoming, normal))) *
           (1 - F) * D * G / (den * den) * fabs_(dot(normal, incoming));  // pow(den, 2)
  }
}
YH_DEV float sample_microfacet_refraction_pdf(float ior, float roughness, f3 normal, f3 outgoing, f3 incoming) {
  refraction_side s = refraction_setup(ior, normal, outgoing);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    f3 halfway = normalize(incoming + outgoing);
    return fresnel_dielectric(s.rel_ior, halfway, outgoing) * sample_microfacet_pdf(roughness, s.up_normal, halfway) /
           (4 * fabs_(dot(outgoing, halfway)));
  } else {
    f3    halfway = -normalize(s.rel_ior * incoming + outgoing) * (s.entering ? 1.0f : -1.0f);
    float den     = s.rel_ior * dot(halfway, incoming) + dot(halfway, outgoing);
    return (1 - fresnel_dielectric(s.rel_ior, halfway, outgoing)) *
           sample_microfacet_pdf(roughness, s.up_normal, halfway) * fabs_(dot(halfway, outgoing)) / (den * den);
  }
}
YH_DEV f3 sample_microfacet_refraction(float ior, float roughness, f3 normal, f3 outgoing, float rnl, float rx,
    float ry) {
  refraction_side s       = refraction_setup(ior, normal, outgoing);
  f3              halfway = sample_microfacet(roughness, s.up_normal, rx, ry);
  if (rnl < fresnel_dielectric(s.entering ? ior : (1 / ior), halfway, outgoing)) return reflect(outgoing, halfway);
  return refract(outgoing, halfway, s.entering ? (1 / ior) : ior);
}

// ---- delta lobes (math.h:4646-4755) ----------------------------------------------
YH_DEV bool ior_is_one(float ior) { return (double)fabs_(ior - 1) < 1e-3; }
YH_DEV f3 eval_delta_reflection(float ior, f3 normal, f3 outgoing, f3 incoming) {
  if (!both_above(normal, outgoing, incoming)) return mk3(0.0f);
  return mk3(1.0f) * fresnel_dielectric(ior, normal, outgoing);
}
YH_DEV f3 eval_delta_reflection(f3 eta, f3 etak, f3 normal, f3 outgoing, f3 incoming) {
  if (!both_above(normal, outgoing, incoming)) return mk3(0.0f);
  return fresnel_conductor(eta, etak, normal, outgoing);
}
YH_DEV float sample_delta_reflection_pdf(f3 normal, f3 outgoing, f3 incoming) {
  return both_above(normal, outgoing, incoming) ? 1.0f : 0.0f;
}
YH_DEV f3 sample_delta_reflection(f3 normal, f3 outgoing) {
  if (dot(normal, outgoing) <= 0) return mk3(0.0f);
  return reflect(outgoing, normal);
}
YH_DEV f3 eval_delta_transmission(f3 normal, f3 outgoing, f3 incoming) {
  if (dot(normal, incoming) >= 0 || dot(normal, outgoing) <= 0) return mk3(0.0f);
  return mk3(1.0f);
}
YH_DEV float sample_delta_transmission_pdf(f3 normal, f3 outgoing, f3 incoming) {
  if (dot(normal, incoming) >= 0 || dot(normal, outgoing) <= 0) return 0;
  return 1;
}
YH_DEV f3 sample_delta_transmission(f3 normal, f3 outgoing) {
  if (dot(normal, outgoing) <= 0) return mk3(0.0f);
  return -outgoing;
}
YH_DEV f3 eval_delta_refraction(float ior, f3 normal, f3 outgoing, f3 incoming) {
  if (ior_is_one(ior)) return dot(normal, incoming) * dot(normal, outgoing) <= 0 ? mk3(1.0f) : mk3(0.0f);
  refraction_side s = refraction_setup(ior, normal, outgoing);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0)
    return mk3(1.0f) * fresnel_dielectric(s.rel_ior, s.up_normal, outgoing);
  return mk3(1.0f) * (1 / (s.rel_ior * s.rel_ior)) * (1 - fresnel_dielectric(s.rel_ior, s.up_normal, outgoing));
}
YH_DEV float sample_delta_refraction_pdf(float ior, f3 normal, f3 outgoing, f3 incoming) {
  if (ior_is_one(ior)) return dot(normal, incoming) * dot(normal, outgoing) < 0 ? 1.0f : 0.0f;
  refraction_side s = refraction_setup(ior, normal, outgoing);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) return fresnel_dielectric(s.rel_ior, s.up_normal, outgoing);
  return (1 - fresnel_dielectric(s.rel_ior, s.up_normal, outgoing));
}
YH_DEV f3 sample_delta_refraction(float ior, f3 normal, f3 outgoing, float rnl) {
  if (ior_is_one(ior)) return -outgoing;
  refraction_side s = refraction_setup(ior, normal, outgoing);
  if (rnl < fresnel_dielectric(s.rel_ior, s.up_normal, outgoing)) return reflect(outgoing, s.up_normal);
  return refract(outgoing, s.up_normal, 1 / s.rel_ior);
}

// ---- the lobe mixture (pt.cpp:372-394, 405-471) ----------------------------------
struct surface_brdf_t {
  f3    diffuse, specular, metal, transmission, refraction;
  float roughness, opacity, ior;
  f3    meta;  // metak is always zero (pt.cpp:440)
  float diffuse_pdf, specular_pdf, metal_pdf, transmission_pdf, refraction_pdf;
};
// color_tex = eval_texture(color_tex, texcoord) and emission_tex_x = eval_texture(emission_tex,
// texcoord, ldr_as_linear).x — the reference scales TRANSMISSION by the emission texture
// (pt.cpp:421-422); both are 1 for an untextured material.
YH_DEV surface_brdf_t surface_brdf(const yhd_material& mat, f3 normal, f3 outgoing, f3 color_tex = {1.0f, 1.0f, 1.0f},
    float emission_tex_x = 1.0f) {
  f3    base         = ld3(mat.color) * color_tex;
  float specular     = mat.specular * 1.0f;
  float metallic     = mat.metallic * 1.0f;
  float roughness    = mat.roughness * 1.0f;
  float transmission = mat.transmission * emission_tex_x;
  bool  thin         = mat.thin || !mat.transmission;
  surface_brdf_t b;
  f3 weight      = mk3(1.0f);
  b.metal        = weight * metallic;
  weight         = weight * (1 - metallic);
  b.refraction   = thin ? mk3(0.0f) : (weight * transmission);
  weight         = weight * (1 - (thin ? 0 : transmission));
  b.specular     = weight * specular;
  weight         = weight * (1 - specular * fresnel_dielectric(mat.ior, outgoing, normal));
  b.transmission = thin ? (weight * transmission * base) : mk3(0.0f);
  weight         = weight * (1 - (thin ? transmission : 0));
  b.diffuse      = weight * base;
  if (mat.color_tex >= 0) {  // reflectivity_to_eta(base) (math.h:4270-4273) of the textured colour
    f3 r   = {fclamp(base.x, 0.0f, 0.99f), fclamp(base.y, 0.0f, 0.99f), fclamp(base.z, 0.0f, 0.99f)};
    b.meta = {(1 + sqrtf(r.x)) / (1 - sqrtf(r.x)), (1 + sqrtf(r.y)) / (1 - sqrtf(r.y)), (1 + sqrtf(r.z)) / (1 - sqrtf(r.z))};
  } else {
    b.meta = ld3(mat.meta);  // the same, precomputed on the host
  }
  b.roughness    = roughness * roughness;
  b.ior          = mat.ior;
  b.opacity      = mat.opacity;    // > 0.999 already snapped to 1 on the host
  if (!is_zero(b.diffuse) || b.roughness) b.roughness = fclamp(b.roughness, 0.03f * 0.03f, 1.0f);
  if (is_zero(b.specular) && is_zero(b.metal) && is_zero(b.transmission) && is_zero(b.refraction)) b.roughness = 1;
  b.diffuse_pdf      = hmax(b.diffuse);
  b.specular_pdf     = hmax(b.specular * fresnel_dielectric(b.ior, normal, outgoing));
  b.metal_pdf        = hmax(b.metal * fresnel_conductor(b.meta, mk3(0.0f), normal, outgoing));
  b.transmission_pdf = hmax(b.transmission);
  b.refraction_pdf   = hmax(b.refraction);
  float pdf_sum = b.diffuse_pdf + b.specular_pdf + b.metal_pdf + b.transmission_pdf + b.refraction_pdf;
  if (pdf_sum) {
    b.diffuse_pdf /= pdf_sum, b.specular_pdf /= pdf_sum, b.metal_pdf /= pdf_sum;
    b.transmission_pdf /= pdf_sum, b.refraction_pdf /= pdf_sum;
  }
  return b;
}
YH_DEV bool is_delta(const surface_brdf_t& b) { return !b.roughness; }  // pt.cpp:495

// eval_brdfcos + sample_brdfcos_pdf (pt.cpp:1069-1102, 1217-1256), rough lobes
YH_DEV void surface_eval_pdf(const surface_brdf_t& b, f3 normal, f3 outgoing, f3 incoming, f3& brdfcos, float& pdf) {
  brdfcos = mk3(0.0f), pdf = 0.0f;
  if (!b.roughness) return;
  if (!is_zero(b.diffuse)) brdfcos = brdfcos + b.diffuse * eval_diffuse_reflection(normal, outgoing, incoming);
  if (!is_zero(b.specular))
    brdfcos = brdfcos + b.specular * eval_microfacet_reflection(b.ior, b.roughness, normal, outgoing, incoming);
  if (!is_zero(b.metal))
    brdfcos = brdfcos + b.metal * eval_microfacet_reflection(b.meta, mk3(0.0f), b.roughness, normal, outgoing, incoming);
  if (!is_zero(b.transmission))
    brdfcos = brdfcos + b.transmission * eval_microfacet_transmission(b.roughness, normal, outgoing, incoming);
  if (!is_zero(b.refraction))
    brdfcos = brdfcos + b.refraction * eval_microfacet_refraction(b.ior, b.roughness, normal, outgoing, incoming);
  if (b.diffuse_pdf) pdf += b.diffuse_pdf * sample_diffuse_reflection_pdf(normal, outgoing, incoming);
  if (b.specular_pdf && !b.refraction_pdf)
    pdf += b.specular_pdf * sample_microfacet_reflection_pdf(b.roughness, normal, outgoing, incoming);
  if (b.metal_pdf) pdf += b.metal_pdf * sample_microfacet_reflection_pdf(b.roughness, normal, outgoing, incoming);
  if (b.transmission_pdf)
    pdf += b.transmission_pdf * sample_microfacet_transmission_pdf(b.roughness, normal, outgoing, incoming);
  if (b.refraction_pdf)
    pdf += b.refraction_pdf * sample_microfacet_refraction_pdf(b.ior, b.roughness, normal, outgoing, incoming);
}
// sample_brdfcos (pt.cpp:1131-1175)
YH_DEV f3 surface_sample(const surface_brdf_t& b, f3 normal, f3 outgoing, float rnl, float rx, float ry) {
  if (!b.roughness) return mk3(0.0f);
  float cdf = 0.0f;
  if (b.diffuse_pdf) {
    cdf += b.diffuse_pdf;
    if (rnl < cdf) return sample_diffuse_reflection(normal, outgoing, rx, ry);
  }
  if (b.specular_pdf && !b.refraction_pdf) {
    cdf += b.specular_pdf;
    if (rnl < cdf) return sample_microfacet_reflection(b.roughness, normal, outgoing, rx, ry);
  }
  if (b.metal_pdf) {
    cdf += b.metal_pdf;
    if (rnl < cdf) return sample_microfacet_reflection(b.roughness, normal, outgoing, rx, ry);
  }
  if (b.transmission_pdf) {
    cdf += b.transmission_pdf;
    if (rnl < cdf) return sample_microfacet_transmission(b.roughness, normal, outgoing, rx, ry);
  }
  if (b.refraction_pdf) {
    cdf += b.refraction_pdf;
    if (rnl < cdf) return sample_microfacet_refraction(b.ior, b.roughness, normal, outgoing, rnl, rx, ry);
  }
  return mk3(0.0f);
}
// eval_delta + sample_delta_pdf (pt.cpp:1104-1128, 1258-1280)
YH_DEV void surface_eval_pdf_delta(const surface_brdf_t& b, f3 normal, f3 outgoing, f3 incoming, f3& brdfcos,
    float& pdf) {
  brdfcos = mk3(0.0f), pdf = 0.0f;
  if (b.roughness) return;
  if (!is_zero(b.specular) && is_zero(b.refraction))
    brdfcos = brdfcos + b.specular * eval_delta_reflection(b.ior, normal, outgoing, incoming);
  if (!is_zero(b.metal)) brdfcos = brdfcos + b.metal * eval_delta_reflection(b.meta, mk3(0.0f), normal, outgoing, incoming);
  if (!is_zero(b.transmission)) brdfcos = brdfcos + b.transmission * eval_delta_transmission(normal, outgoing, incoming);
  if (!is_zero(b.refraction)) brdfcos = brdfcos + b.refraction * eval_delta_refraction(b.ior, normal, outgoing, incoming);
  if (b.specular_pdf && !b.refraction_pdf) pdf += b.specular_pdf * sample_delta_reflection_pdf(normal, outgoing, incoming);
  if (b.metal_pdf) pdf += b.metal_pdf * sample_delta_reflection_pdf(normal, outgoing, incoming);
  if (b.transmission_pdf) pdf += b.transmission_pdf * sample_delta_transmission_pdf(normal, outgoing, incoming);
  if (b.refraction_pdf) pdf += b.refraction_pdf * sample_delta_refraction_pdf(b.ior, normal, outgoing, incoming);
}
// sample_delta (pt.cpp:1177-1214)
YH_DEV f3 surface_sample_delta(const surface_brdf_t& b, f3 normal, f3 outgoing, float rnl) {
  if (b.roughness) return mk3(0.0f);
  float cdf = 0.0f;
  cdf += b.diffuse_pdf;
  if (b.specular_pdf && !b.refraction_pdf) {
    cdf += b.specular_pdf;
    if (rnl < cdf) return sample_delta_reflection(normal, outgoing);
  }
  if (b.metal_pdf) {
    cdf += b.metal_pdf;
    if (rnl < cdf) return sample_delta_reflection(normal, outgoing);
  }
  if (b.transmission_pdf) {
    cdf += b.transmission_pdf;
    if (rnl < cdf) return sample_delta_transmission(normal, outgoing);
  }
  if (b.refraction_pdf) {
    cdf += b.refraction_pdf;
    if (rnl < cdf) return sample_delta_refraction(b.ior, normal, outgoing, rnl);
  }
  return mk3(0.0f);
}

// ---- colour textures (pt.cpp:167-200 on texels converted at upload, yh_device.h) ---------
YH_DEV f3 eval_texture(const yhd_scene& sc, int tex, bool ldr_as_linear, float u_, float v_) {
  if (tex < 0) return mk3(1.0f);
  const yhd_texture& t  = sc.textures[tex];
  int                sx = t.width, sy = t.height;
  float s = fmodf(u_, 1.0f) * sx;
  if (s < 0) s += sx;
  float tt = fmodf(v_, 1.0f) * sy;
  if (tt < 0) tt += sy;
  int   i = iclamp((int)s, 0, sx - 1), j = iclamp((int)tt, 0, sy - 1);
  int   ii = (i + 1) % sx, jj = (j + 1) % sy;
  float u = s - i, v = tt - j;
  const yhd_float4* tx = sc.tex_texels + (ldr_as_linear ? t.linear_base : t.srgb_base);
  f3 a = xyz(tx[(size_t)j * sx + i]), b = xyz(tx[(size_t)jj * sx + i]);
  f3 c = xyz(tx[(size_t)j * sx + ii]), d = xyz(tx[(size_t)jj * sx + ii]);
  return a * (1 - u) * (1 - v) + b * (1 - u) * v + c * u * (1 - v) + d * u * v;
}

// ---- homogeneous volumes (math.h:4758-4822, pt.cpp:1360-1377) ---------------------
struct vsdf_t {
  f3    density, scatter;
  float anisotropy;
};
YH_DEV f3 exp3(f3 a) { return {expf(a.x), expf(a.y), expf(a.z)}; }
YH_DEV f3 eval_transmittance(f3 density, float distance) { return exp3(-density * distance); }
YH_DEV float sample_transmittance(f3 density, float max_distance, float rl, float rd) {
  int   channel  = iclamp((int)(rl * 3), 0, 2);
  float d        = channel == 0 ? density.x : (channel == 1 ? density.y : density.z);
  float distance = (d == 0) ? flt_max : -logf(1 - rd) / d;
  return fmin_(distance, max_distance);
}
YH_DEV float sample_transmittance_pdf(f3 density, float distance, float max_distance) {
  if (distance < max_distance) {
    f3 t = density * exp3(-density * distance);
    return (t.x + t.y + t.z) / 3;
  }
  f3 t = exp3(-density * max_distance);
  return (t.x + t.y + t.z) / 3;
}
YH_DEV float eval_phasefunction(float anisotropy, f3 outgoing, f3 incoming) {
  float cosine = -dot(outgoing, incoming);
  float denom  = 1 + anisotropy * anisotropy - 2 * anisotropy * cosine;
  return (1 - anisotropy * anisotropy) / (4 * pif * denom * sqrtf(denom));
}
YH_DEV f3 sample_phasefunction(float anisotropy, f3 outgoing, float rx, float ry) {
  float cos_theta;
  if (fabs_(anisotropy) < 1e-3f) {
    cos_theta = 1 - 2 * ry;
  } else {
    float square = (1 - anisotropy * anisotropy) / (1 + anisotropy - 2 * anisotropy * ry);
    cos_theta    = (1 + anisotropy * anisotropy - square * square) / (2 * anisotropy);
  }
  float sin_theta = sqrtf(fmax_(0.0f, 1 - cos_theta * cos_theta));
  float phi       = 2 * pif * rx;
  f3    local     = {sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta};
  // basis_fromz(-outgoing) * local: a plain matrix product, not normalised (math.h:4815)
  f3    zz   = normalize(-outgoing);
  float sign = copysignf(1.0f, zz.z);
  float a    = -1.0f / (sign + zz.z);
  float b    = zz.x * zz.y * a;
  f3    x    = {1.0f + sign * zz.x * zz.x * a, sign * b, -sign * zz.x};
  f3    y    = {b, sign + zz.y * zz.y * a, -zz.y};
  return x * local.x + y * local.y + zz * local.z;
}
YH_DEV f3 eval_scattering(const vsdf_t& v, f3 outgoing, f3 incoming) {
  if (is_zero(v.density)) return mk3(0.0f);
  return v.scatter * v.density * eval_phasefunction(v.anisotropy, outgoing, incoming);
}
YH_DEV f3 sample_scattering(const vsdf_t& v, f3 outgoing, float rx, float ry) {
  if (is_zero(v.density)) return mk3(0.0f);
  return sample_phasefunction(v.anisotropy, outgoing, rx, ry);
}
YH_DEV float sample_scattering_pdf(const vsdf_t& v, f3 outgoing, f3 incoming) {
  if (is_zero(v.density)) return 0;
  return eval_phasefunction(v.anisotropy, outgoing, incoming);
}

}  // namespace yhd
#endif
