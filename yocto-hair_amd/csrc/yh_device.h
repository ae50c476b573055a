// yh_device.h — layout of the scene and render state in HBM, shared by the
// host upload code (g++) and the HIP kernels (hipcc). Plain C structs only.
//
// Everything a ray touches is a 16-byte-aligned record fetched with dwordx4
// loads, laid out so that one traversal step is ONE dependent fetch:
//   * shape BVH node (128 B, one cache line): the reference's binary tree
//     (pt.cpp:557-650) with two levels collapsed into a 4-wide node, SoA over
//     the four slots: minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4] ref[4]
//     {axes,0,0,0} (host/bvh_build.h: WideNode; WideNode8 = three levels, 256 B). Wide nodes are numbered
//     breadth-first, so the first K nodes are the top of the tree.
//   * scene-level node (32 B): {min.xyz, start} {max.xyz, meta}; meta = num |
//     internal << 16 | axis << 24 — the reference's bvh_node (pt.h:243-249).
//   * hair segment (64 B), stored in BVH LEAF ORDER (no primitives[] / lines[]
//     / positions[] / radius[] indirection): {p0.xyz, r0} {p1.xyz, r1} used by
//     the ray test, then {t0.xyz, element} {t1.xyz, 0} (vertex tangents) used
//     only when the hit is shaded — no index chasing after a hit either.
//   * triangle (96 B) in leaf order: {p0.xyz, element} {p1.xyz,0} {p2.xyz,0}
//     {n0.xyz,0} {n1.xyz,0} {n2.xyz,0}.
//   * per-vertex / per-element arrays (vpos, elems) are kept only for sampling
//     a point on an area light by triangle index (pt.cpp:1288-1290).
#ifndef YH_DEVICE_H_
#define YH_DEVICE_H_
#include <stdint.h>

#define YH_KIND_LINES 1
#define YH_KIND_TRIANGLES 2
#define YH_TILE 8          /* tiles are 8x8 pixels = one 64-lane wavefront  */
#define YH_MAX_LIGHTS 16
#define YH_MAX_ENVS 4

typedef struct yhd_float4 { float x, y, z, w; } yhd_float4;
typedef struct yhd_int4 { int x, y, z, w; } yhd_int4;

// One instance (ptr::object) with its shape's array offsets folded in.
typedef struct yhd_object {
  float frame[12];      // object -> world
  float inv_frame[12];  // inverse(frame, non_rigid = true), pt.cpp:1012-1013
  int   kind;           // YH_KIND_*
  int   node_base;      // (unused since round 6: the trees exist only in lane_blob; kept so that the LDS copy of the record keeps its layout)
  int   prim_base;      // first leaf-ordered record of the shape (float4 units)
  int   vert_base;      // first vertex in vpos / vtex (shapes that need them, host/scene_upload.cpp)
  int   elem_base;      // first element in elems
  int   has_normals;
  int   material;
  int   has_texcoords;  // the shape has per-vertex texture coordinates (else texcoord = element uv)
  // world-space box of the object (transform_bbox of the shape's root box, pt.cpp:806) grown by a
  // safety margin: a ray that misses it by that much cannot hit anything in the object, so ENTER
  // and the root fetch are skipped for it (dev_trace.h)
  float wbox_min[4], wbox_max[4];  // ([3]: unused)
  // the shape's place in yhd_scene::lane_blob (the one-lane kernels' copy of the trees, dev_lane.h), in 32-byte units:
  // its root node and its first test record
  int   lane_root, lane_test;
  int   lane_root8, lane_root16;  // ... and the roots of its 8- / 16-wide trees there
} yhd_object;
#define YH_OBJECT_F4 11 /* sizeof(yhd_object) / 16 */

// ptr::material + everything of hair_brdf that depends on the material only
// (ext.cpp:131-172), computed ONCE on the host at upload.
typedef struct yhd_material {
  float  emission[3];
  float  color[3];
  float  diffuse_pdf;  // 1 if any colour channel is non-zero, else 0
  int    thin;
  // hair constants
  float  sigma_a[3];
  float  alpha, eta;
  float  v[4];
  float  s;
  float  sin_2k_alpha[3], cos_2k_alpha[3];
  // derived per-lobe constants used by mp() / np() / sampling
  float  inv_v[4];         // 1 / v[p]
  float  log_inv_2v[4];    // log(1 / (2 v[p]))
  float  exp_m2_inv_v[4];  // exp(-2 / v[p])
  double mp_den[4];        // sinh(1 / v[p]) * 2 * v[p]   (double, ext.cpp:206)
  float  tl_cdf_a;         // logistic_cdf(-pi, s)
  float  tl_norm;          // logistic_cdf(pi, s) - logistic_cdf(-pi, s)
  // surface lobes other than diffuse (dev_surface.h). `plain` = none of them:
  // specular = metallic = transmission = 0 and opacity = 1, the only kind of
  // surface in the BASELINE configs, which keeps its short diffuse-only path.
  int    plain;
  float  specular, metallic, roughness, ior, transmission;
  float  opacity;          // material opacity, snapped to 1 above 0.999 (pt.cpp:454)
  float  meta[3];          // reflectivity_to_eta(color) (pt.cpp:439)
  // the medium inside a closed transmissive object (eval_vsdf, pt.cpp:504-527)
  int    has_volume;       // !thin && transmission (pt.cpp:531)
  float  vol_density[3];   // -log(clamp(color, 1e-4, 1)) / trdepth
  float  vol_scatter[3];
  float  vol_anisotropy;
  // colour textures (index into yhd_scene::textures, -1 = none); trdepth for the textured medium
  int    emission_tex, color_tex, scattering_tex;
  float  trdepth;
} yhd_material;

// A colour texture: float4 texels in yhd_scene::tex_texels, already converted per texel the way
// lookup_texture (pt.cpp:147-164) does. `srgb_base`: byte textures decoded sRGB -> linear (float
// textures as they are); `linear_base`: byte / 255 without decoding (only the transmission
// factor reads it, pt.cpp:421-422), -1 when no material needs it.
typedef struct yhd_texture {
  int width, height;
  int srgb_base, linear_base;
} yhd_texture;

typedef struct yhd_light {
  int object;       // >= 0: area light on that object
  int environment;  // >= 0: environment light
  int cdf_base;     // offset into light_cdf
  int cdf_count;    // triangles / texels (0 for a constant environment)
  int small_base;   // >= 0: an area light of at most YH_SMALL_LIGHT_TRIS triangles, its record at this float4 of
                    // yhd_scene::light_table (staged in LDS by the kernels); -1: sampled / intersected through memory
} yhd_light;
// Record of a small area light (yhd_scene::light_table): the shading code samples it and intersects the light-pdf
// rays with it (sample_lights / sample_lights_pdf, pt.cpp:1283-1358) without a memory access — through the BVH
// that is six dependent fetches per bounce for a two-triangle light.
//   [0] = {root box min.xyz, number of triangles (int)}   [1] = {root box max.xyz, total area}
//   [2 + 3 i ..] = triangle i in LEAF order: {p0.xyz, element (int)} {p1.xyz, 0} {p2.xyz, 0}
//   [14] = area cdf by ELEMENT (light_cdf's entries)
#define YH_SMALL_LIGHT_TRIS 4
#define YH_SMALL_LIGHT_F4 15

typedef struct yhd_environment {
  float frame[12];
  float inv_frame[12];  // rigid inverse (transpose), pt.cpp:539
  float emission[3];
  int   tex_w, tex_h;   // 0 when constant
  int   texel_base;     // offset into env_texels (float4 per texel)
  int   pad0, pad1;
} yhd_environment;

typedef struct yhd_camera {
  float frame[12];
  float lens, film_x, film_y, focus, aperture;
} yhd_camera;

typedef struct yhd_scene {
  // geometry
  const yhd_float4* nodes;      // NULL since round 6: the node arrays exist only inside lane_blob (below); the field keeps the struct's layout
  const yhd_float4* prims;      // leaf-ordered records (4 or 6 float4 each)
  const yhd_float4* vpos;       // per vertex {pos, radius}
  const yhd_int4*   elems;      // per element vertex indices (shape-local)
  const yhd_object* objects;
  const yhd_material* materials;
  // scene-level BVH over instances
  const yhd_float4* scene_nodes;
  const int*        scene_prims;
  int               num_scene_nodes;
  int               num_objects;
  // lights
  int               num_lights;
  int               num_environments;
  yhd_light         lights[YH_MAX_LIGHTS];
  yhd_environment   environments[YH_MAX_ENVS];
  const float*      light_cdf;
  const yhd_float4* env_texels;
  const yhd_texture* textures;   // material colour textures
  const yhd_float4* tex_texels;
  const float*      vtex;        // per vertex texture coordinates (2 floats), indexed like vpos
  yhd_camera        camera;
  int               num_nodes_total; // wide nodes the 4-wide trees hold
  int               num_prim_f4;     // float4 in `prims`
  int               lds_scene_f4;       // float4 count of the scene-level LDS table (0: scene too big, read from memory)
  int               general_materials;  // some material has lobes beyond diffuse / hair, or some area light is not a small one: the GENERAL kernel variants
  int               stack_entries;   // traversal stack depth per ray the kernels reserve in LDS (>= the trees' need)
  // tables the kernels stage in LDS next to the scene-level table (dev_trace.h: stage_tables)
  const yhd_float4* light_table;     // small area lights, YH_SMALL_LIGHT_F4 float4 each
  int               light_table_f4;
  // coarse index of ONE environment light's texel cdf: env_tab[k] = cdf[min(n, (k + 1) * env_tab_stride) - 1], so that
  // the first log2(env_tab_k) steps of the binary search of sample_lights (math.h:4957-4962, 21 dependent
  // fetches for sky.hdr) read LDS
  const float*      env_tab;
  int               env_tab_light;   // index into lights[], -1: none
  int               env_tab_k, env_tab_stride;
  int               lds_materials;   // materials staged in LDS (all of them, or 0 when they are too many)
  // the same trees with THREE binary levels per node (host/bvh_build.h: WideNode8, 256 B: YH_MODE_OCT, dev_trace.h) and with FOUR
  // (WideNode16, 512 B: YH_MODE_HEX) live in the lane blob below, behind the 4-wide nodes; traversal stack depths per ray over them
  // (up to 7 / 15 pushes per node):
  int               stack_entries8;
  int               stack_entries16;
  // THE TREES AS EVERY TRAVERSAL KERNEL READS THEM: ONE array addressed in 32-byte units, so that whatever a lane, a quad, an octet or a
  // group of sixteen holds — node or leaf — is fetched from lane_blob + 32 * offset:
  //   test records  a hair segment = 32 B {p0.xyz, r0} {p1.xyz, r1} (the ray-test half of its yhd_scene::prims record),
  //                 a triangle = 64 B {p0}{p1}{p2}{-}; both in the shape's leaf order
  //   nodes         4-wide (128 B: two levels of the reference's binary tree; the axes word + a bit per occupied slot in bits 8-11), then
  //                 8-wide (256 B, three levels), then 16-wide (512 B, four levels); slot = {min.xyz, max.x}{max.yz, ref, axes};
  //                 references ABSOLUTE: a child node's first slot in the blob, or YH_TAG_LEAF | count << 27 | the leaf's first test record
  // Made on the device inside yh_upload_scene (csrc/bvh_gpu.hip: k_wide_collapse; csrc/stream.hip: k_lane_tests) from the binary tree and
  // the leaf records; the one-lane kernels read the part up to the end of the 4-wide nodes through 32-bit byte offsets.
  const yhd_float4* lane_blob;
  long long         lane_blob_units;
} yhd_scene;
#define YH_MATERIAL_F4 17 /* sizeof(yhd_material) / 16 */
// float4 the kernels reserve in LDS for the tables: scene level | camera (5) | small lights | env cdf index | materials
#define YHD_LDS_TABLES_F4(sc) ((sc)->lds_scene_f4 + 5 + (sc)->light_table_f4 + ((sc)->env_tab_k + 3) / 4 + YH_MATERIAL_F4 * (sc)->lds_materials)

// Render state (pt.h:419-429) in SoA form.
typedef struct yhd_state {
  uint64_t*   rng_state;  // per pixel
  uint64_t*   rng_inc;    // per pixel
  yhd_float4* accum;      // per pixel: sum of clamped radiance, w = hit count
  // Work items of a launch, in hand-out order: item = tile id * 4 + quadrant
  // (one 4x4-pixel quadrant of an 8x8 tile = 16 quads = one wavefront).
  const int*  tiles;
  int*        tile_cursor;  // next position in `tiles` (zeroed before every launch); k_stream: one cursor per item
                            // group, 16 ints apart
  // k_stream: the hand-out list in `num_groups` groups, group g = tiles[group_begin[g] .. group_begin[g + 1]).
  // Workgroups are placed round-robin over the 8 XCDs, each with an L2 of its own: the workgroups of one XCD
  // (blockIdx % 8) take the items of ONE compact image region, so that what an XCD's L2 holds is the part of the
  // scene behind that region and not an eighth of everything; a group that runs dry takes from the next.
  int         num_groups;
  int         group_begin[9];
  unsigned int* tile_cost;  // per item: wall-clock ticks (100 MHz) its last launch took
  int         num_tiles;  // number of work items in `tiles`
  int         shader;        // YH_SHADER_* (yhair.h): path is the product path, the others preview / debug
  int         launch_shape;  // 0: k_trace 512 threads x 4 waves per SIMD; 1: k_trace 256 x 5 (dense scenes); 3: k_stream; 4-8: the wide forms (host/context_internal.h)
  int         width, height;
  int         tiles_x;
  int         samples_done;
  int         bounces;
  float       clamp;
  int         shard_rank, shard_world;  // tile ids owned: rank, rank + world, ...
  // (k_trace hands the first grid x waves-per-workgroup entries of `tiles` out BY POSITION — wave w of workgroup b starts
  // with entry b * (waves per workgroup) + w, the rest go through the cursor — and the host lays that head of the list out by
  // hardware wave slot, host/launch_plan.cpp: lay_out_first_round.)
} yhd_state;

// Path pool of the streaming integrator (csrc/stream.hip): one lane per path, `slots_per_wave` slots owned by
// each WAVEFRONT (no workgroup-level synchronisation), SoA over the slots.
// One path slot = one 128-byte cache line: a stage touches one line per path, not one per field. The fields are grouped by WRITER into the line's
// four 32-byte sectors (round 6; csrc/stream.hip: SLOT_*), so that a stage dirties the sectors it owns and not the line.
typedef struct yhd_path_slot {
  yhd_float4 ray_o;     // sector 0: origin.xyz
  yhd_float4 ray_d;     //           direction.xyz
  yhd_float4 weight;    // sector 1: path weight.xyz; w = int bits: bounce | hit << 8 | in_medium << 9
  yhd_int4   rngw;      //           the pixel's PCG32 state lo, hi; traversal steps of the pixel's rays so far (scheduling hint)
  yhd_int4   hit;       // sector 2: object (-1 miss, -2 path ended in shading, -3 new pixel), leaf slot, u bits, v bits
  yhd_int4   hit2;      //           distance bits of the closest hit, steps of the ray just traced
  yhd_float4 radiance;  // sector 3: radiance collected so far .xyz; w = int bits: the pixel
  yhd_int4   own;       //           samples left to start, work item, the stream's inc lo, hi
} yhd_path_slot;
typedef struct yhd_stream {
  yhd_path_slot* slots;
  yhd_float4* medium;    // scenes with volumes only, 2 per slot
  unsigned int* stack_ovf;  // per wave: ovf_entries x 64 lanes, what the LDS stack window spills (dev_lane.h)
  unsigned long long* prof;    // developer build (YHAIR_ST_PROF): 32 counters, see stream.hip; else NULL
  unsigned long long* wave_log;  // per wave {begin, end} of its work in ticks of the 100 MHz wall clock; NULL = not kept
  // A wave's OWN share of the work list (round 5): wave w starts with the entries [wave_begin[w], wave_begin[w + 1]) of yhd_state::tiles
  // (a multiple of four entries, padded with -1 = no item: a take is four entries = 64 pixels)
  // — the host sizes the shares by how fast the wave's hardware slot runs (host/launch_plan.cpp: deal_items_for_stream) — and takes from
  // the cursor (the entries behind the shares: yhd_state::group_begin[0]) only after them. NULL: everything through the cursor.
  const int*  wave_begin;
  int*        wave_fill;   // per wave: slots [0, wave_fill[w]) of its pool hold the pixels of its own share (k_stream_seed writes, k_stream reads); NULL with wave_begin
  int         slots_per_wave;  // multiple of 64, <= 4096
  int         ovf_entries;
  long long   total_slots;     // slots in the pool (all waves)
  int         prof_parts_only;  // developer build (YHAIR_ST_PROF=2): only the five time stamps per step, none of the per-branch counters (they lengthen the step they measure)
  int         suspend_lanes;   // a wave whose ray list is dry leaves the trace stage for the shading stages when at most this many of its lanes are busy (< 64; stream.hip)
} yhd_stream;

// Work counters (one 64-bit slot each), accumulated with atomics by the
// instrumented kernel variant only.
typedef struct yhd_counters {
  unsigned long long samples, rays, nodes, seg, tri, hair, surf, envl, envs;
  // wave-level profile of the instrumented build (developer diagnostics):
  // shader-clock cycles inside trace_ray / path_step, wave-level iterations of
  // the regeneration loop, and traversal-loop trip counts (per wave = max over
  // lanes; per lane = sum over lanes) of the main rays
  unsigned long long cyc_trace, cyc_shade, cyc_tile, wave_iters, wave_steps, lane_steps, lane_iters;
  unsigned long long c_geom, c_sample, c_eval, c_rest;  // lane-0 cycles inside path_step: hit geometry, direction sampling, BSDF eval+pdf, the rest
  // traversal divergence: (wave trips, active lanes) for node, line-leaf, triangle-leaf, ENTER, scene-node code
  unsigned long long branch[10];
} yhd_counters;

#endif
