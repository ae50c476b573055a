// dev_math.h — device-side vector kit for the hair path (gfx950).
//
// The kernels are compiled with -ffp-contract=off: every +,-,*,/ and sqrt is
// a single correctly-rounded IEEE operation in the order the reference writes
// it (libs/yocto/yocto_math.h:1778-1812,1979-2113), so ray generation, BVH
// traversal and the ray-line / ray-triangle tests are BIT-IDENTICAL to the
// reference's CPU results; only libm-type functions (sin, cos, exp, log,
// atan2, acos) may differ in the last ulp.
#ifndef YH_DEV_MATH_H_
#define YH_DEV_MATH_H_
#include <hip/hip_runtime.h>

#include "yh_device.h"

#define YH_DEV __device__ __forceinline__

namespace yhd {

constexpr float pif     = 3.14159274101257324f;  // (float)pi, math.h:214-215
constexpr float flt_max = 3.402823466e+38f;
constexpr float ray_eps = 1e-4f;                 // math.h:1106

struct f3 {
  float x, y, z;
};
YH_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
YH_DEV f3 mk3(float a) { return f3{a, a, a}; }
YH_DEV f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
YH_DEV f3 xyz(const yhd_float4& a) { return f3{a.x, a.y, a.z}; }
YH_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
YH_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
YH_DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
YH_DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
YH_DEV f3 operator*(f3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
YH_DEV f3 operator*(float a, f3 b) { return {a * b.x, a * b.y, a * b.z}; }
YH_DEV f3 operator/(f3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
YH_DEV f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
YH_DEV f3 operator-(float a, f3 b) { return {a - b.x, a - b.y, a - b.z}; }

// math.h:1778-1783: comparisons written exactly like the reference so that
// NaN operands take the same branch.
YH_DEV float fabs_(float a) { return a < 0 ? -a : a; }
YH_DEV float fmin_(float a, float b) { return (a < b) ? a : b; }
YH_DEV float fmax_(float a, float b) { return (a > b) ? a : b; }
YH_DEV float fclamp(float a, float lo, float hi) { return fmin_(fmax_(a, lo), hi); }
YH_DEV int   iclamp(int a, int lo, int hi) { return min(max(a, lo), hi); }

YH_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
YH_DEV f3    cross(f3 a, f3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
YH_DEV float length(f3 a) { return sqrtf(dot(a, a)); }
YH_DEV f3    normalize(f3 a) {  // math.h:2036-2039
  float l = length(a);
  return (l != 0) ? a / l : a;
}
YH_DEV f3    orthonormalize(f3 a, f3 b) { return normalize(a - b * dot(a, b)); }
YH_DEV float hmax(f3 a) { return fmax_(fmax_(a.x, a.y), a.z); }
YH_DEV float hmin(f3 a) { return fmin_(fmin_(a.x, a.y), a.z); }
YH_DEV bool  is_zero(f3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
YH_DEV bool  finite3(f3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }
YH_DEV float luminance(f3 a) { return (0.2126f * a.x + 0.7152f * a.y + 0.0722f * a.z); }

// frames are 12 floats: x, y, z, o (math.h:932-936)
struct frame {
  f3 x, y, z, o;
};
YH_DEV frame ldframe(const float* f) {
  return frame{ld3(f), ld3(f + 3), ld3(f + 6), ld3(f + 9)};
}
// math.h:3136-3144
YH_DEV f3 transform_point(const frame& a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.o; }
YH_DEV f3 transform_vector(const frame& a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
YH_DEV f3 transform_direction(const frame& a, f3 b) { return normalize(transform_vector(a, b)); }
// rigid inverse of a rotation-only frame = transpose (math.h:2881-2884)
YH_DEV frame transpose_rot(const frame& a) {
  frame r;
  r.x = {a.x.x, a.y.x, a.z.x};
  r.y = {a.x.y, a.y.y, a.z.y};
  r.z = {a.x.z, a.y.z, a.z.z};
  r.o = -(r.x * a.o.x + r.y * a.o.y + r.z * a.o.z);
  return r;
}

// ---------------------------------------------------------------------------
// Quads: four adjacent lanes that own one ray / one pixel (dev_trace.h).
// ---------------------------------------------------------------------------
// DPP quad permutes: data exchange between the four lanes of a quad in one
// VALU instruction (quad_perm control = sel0 | sel1 << 2 | sel2 << 4 | sel3 << 6)
template <int CTRL>
YH_DEV int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
YH_DEV float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int K>
YH_DEV unsigned int quad_bcast_u(unsigned int v) { return (unsigned int)dpp_i<K * 0x55>((int)v); }
template <int K>
YH_DEV float quad_bcast_f(float v) { return dpp_f<K * 0x55>(v); }
#define YH_QUAD_XOR1 0xB1 /* quad_perm [1,0,3,2] */
#define YH_QUAD_XOR2 0x4E /* quad_perm [2,3,0,1] */
// The four lanes of a quad run the same path, so an IEEE division (ten instructions: scale, rcp,
// three fma refinements, fmas, fixup) of each component of a vector is computed four times over.
// These forms give component q to lane q (lane 3 repeats component 2) and broadcast the three
// quotients: one division per lane instead of three, the same instruction on the same operands, so
// the same bits. ONLY where all four lanes of the quad are active and hold identical operands.
YH_DEV float quad_pick(f3 a) {
  unsigned int q = __lane_id() & 3u;
  return q == 0 ? a.x : (q == 1 ? a.y : a.z);
}
YH_DEV f3 quad_spread(float r) { return f3{quad_bcast_f<0>(r), quad_bcast_f<1>(r), quad_bcast_f<2>(r)}; }
// YH_LANE = 1 (csrc/stream.hip): ONE LANE PER PATH instead of a quad. The quad_* forms below become the
// plain per-lane expressions (same operations on the same operands, so the same bits), and the
// callers pick the lane forms of the few functions that exchange data inside a quad (dev_path.h).
#ifndef YH_LANE
#define YH_LANE 0
#endif
YH_DEV f3 quad_div(f3 a, float b) { return !YH_LANE ? quad_spread(quad_pick(a) / b) : a / b; }
YH_DEV f3 quad_rcp(f3 b) { return !YH_LANE ? quad_spread(1 / quad_pick(b)) : f3{1 / b.x, 1 / b.y, 1 / b.z}; }
YH_DEV f3 quad_normalize(f3 a) {  // normalize (math.h:2036-2039)
  float l = length(a);
  return (l != 0) ? quad_div(a, l) : a;
}
YH_DEV f3 quad_orthonormalize(f3 a, f3 b) { return quad_normalize(a - b * dot(a, b)); }
// 4-bit mask of `p` over the lanes of this lane's quad
YH_DEV unsigned int quad_ballot(bool p) {
  unsigned long long b = __ballot(p);
  return (unsigned int)(b >> (__lane_id() & ~3u)) & 15u;
}

}  // namespace yhd
#endif
