/*
 * yh_oracle.h — TEST INFRASTRUCTURE ONLY. Never linked, imported or executed
 * by the product path (yocto-hair_amd/). Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * CPU restatement (plain scalar C++, g++ -O3, no fast-math) of the hair
 * path-tracing hot path of dsforza96/yocto-hair. Every function in
 * yh_oracle.cpp cites the reference file:line it follows.
 *
 * PARITY PINNED: in the build container this restatement is checked
 * bit-for-bit against the real reference (oracle/_ref/libyh_ref.so, built by
 * oracle/Makefile from /root/reference): RNG streams, hair BSDF
 * brdf/eval/sample/pdf, ray-line / ray-triangle / ray-bbox tests, closest hits
 * and whole rendered images (tests/test_oracle_vs_ref.py), and against the
 * committed golden vectors generated from the reference
 * (tests/golden/, oracle/make_golden.py) everywhere else.
 */
#ifndef YH_ORACLE_H_
#define YH_ORACLE_H_
#include <stdint.h>

#include "yhair.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct yo_scene yo_scene;

/* math.h:1405-1442, pt.cpp:1942-1945 */
void yo_rng_stream(uint64_t seed, uint64_t seq, int n, uint64_t* state_inc,
    float* out);
void yo_pixel_seqs(int n, int* out);

/* hair material = 12 floats: sigma_a[3] beta_m beta_n alpha eta color[3]
 * eumelanin pheomelanin (ext.h:86-95); brdf = 30 floats (yhair.h)          */
void yo_hair_brdf(int n, const float* mats, const float* v, const float* nrm,
    const float* tng, float* out);
void yo_hair_eval(
    int n, const float* brdf, const float* wo, const float* wi, float* out);
void yo_hair_sample(
    int n, const float* brdf, const float* wo, const float* rn, float* out);
void yo_hair_pdf(
    int n, const float* brdf, const float* wo, const float* wi, float* out);
/* ext.cpp:555-693: returns 1 = "OK!", 0 = "TEST FAILED!" */
int yo_selftest(int which, float* worst);

/* yocto_math.h:4427-4755: one lobe (YH_LOBE_*); params 8n (ior, roughness,
 * eta[3], etak[3]); rn 3n (rnl, rn.x, rn.y); out 7n (f*|cos| [3], pdf,
 * sampled incoming [3])                                                     */
void yo_surface_lobe(int kind, int n, const float* params, const float* normal,
    const float* outgoing, const float* incoming, const float* rn, float* out);
/* pbrt curve -> 5-vertex strand (yocto_pbrt.h:1751-1797): P 12n; out positions 15n,
 * normals 15n, radius 5n, lines 8n (indices from base_vertex + 5 * curve)       */
void yo_curves_to_lines(int n, const float* P, const float* width0, const float* width1,
    int base_vertex, float* positions, float* normals, float* radius, int* lines);
/* fresnel_dielectric(ior), fresnel_conductor(eta, etak), reflectivity_to_eta(eta): out 7n */
void yo_fresnel(int n, const float* params, const float* normal, const float* outgoing, float* out);
/* pt.cpp:405-471 (eval_brdf) + 1069-1280 (dispatch), non-hair materials      */
void yo_surface_bsdf(int n, const yh_material* materials, const float* normal,
    const float* outgoing, const float* incoming, const float* rn, float* out);

/* math.h:3426-3505,3544-3554 */
void yo_intersect_line(int n, const float* rays, const float* p0,
    const float* p1, const float* r0, const float* r1, int* hit, float* uv,
    float* dist);
void yo_intersect_triangle(int n, const float* rays, const float* p0,
    const float* p1, const float* p2, int* hit, float* uv, float* dist);
void yo_intersect_bbox(int n, const float* rays, const float* bbox, int* hit);

/* init_bvh + init_lights (pt.cpp:755-818,1695-1740) on a copy of the scene */
yo_scene* yo_scene_create(const yh_scene_desc* desc);
void      yo_scene_free(yo_scene* scene);
int       yo_scene_num_lights(const yo_scene* scene);
/* pt.cpp:1039-1046; element/object -1 on miss */
void yo_scene_intersect(const yo_scene* scene, int n, const float* rays,
    int* object, int* element, float* uv, float* dist);
void yo_scene_intersect_counted(const yo_scene* scene, int n, const float* rays, int* nodes, int* prims);
/* BVH export for structural checks: returns node count of shape `shape`
 * (-1 = scene-level BVH); fills nodes (8 floats: bbox min/max, then start,
 * num|internal<<16|axis<<24 as int bits) and primitives when non-NULL.       */
int yo_scene_bvh(const yo_scene* scene, int shape, float* nodes, int* prims);

/* init_state + samples x trace_samples (pt.cpp:1931-2007). nthreads <= 0:
 * hardware_concurrency(). rgba (W*H*4), rng (W*H*2 u64) and counts may be
 * NULL. If rgba is NULL only the size is returned.                          */
int yo_render(const yo_scene* scene, const yh_trace_params* params,
    int samples, int nthreads, int* width, int* height, float* rgba,
    uint64_t* rng_state_inc, yh_workcounts* counts);

#ifdef __cplusplus
}
#endif
#endif
