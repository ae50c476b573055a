/*
 * yhair.h — C ABI of the MI355X-native hair path-tracing sample loop.
 *
 * This is the drop-in boundary for the ONE hot path of dsforza96/yocto-hair:
 *   trace_samples -> trace_sample -> trace_path -> {BVH traversal, ray-line /
 *   ray-triangle intersection, hair BSDF eval / sample / pdf, light MIS}.
 *
 * The reference has no FFI layer. The narrowest seam it has is the
 * yocto::pathtrace C++ API (libs/yocto_pathtrace/yocto_pathtrace.h:207-230:
 * init_bvh, init_lights, init_state, trace_samples) and, one level down, the
 * four yocto::extension functions (libs/yocto_extension/yocto_extension.h:
 * 115-125). Every entry point below names the reference interface it replaces.
 *
 * Conventions: plain C types only, no exceptions cross the boundary. Every
 * function returning int returns YH_OK (0) on success or a negative YH_E_*
 * code; yh_last_error() gives the text. Host arrays passed in are borrowed for
 * the duration of the call only. A context owns all of its device memory and
 * is bound to ONE GPU (one process per GPU is the deployment model; image
 * tiles are sharded across processes with yh_set_shard()).
 *
 * All vectors are packed floats; frames are 12 floats x,y,z,o column vectors
 * exactly as yocto::math::frame3f (libs/yocto/yocto_math.h, frame3f).
 */
#ifndef YHAIR_H_
#define YHAIR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YH_OK 0
#define YH_E_INVALID -1     /* bad argument / unsupported scene feature      */
#define YH_E_DEVICE -2      /* HIP runtime error (no GPU, launch failure...) */
#define YH_E_STATE -3       /* call order violated (e.g. trace before init)  */
#define YH_E_IO -4          /* file not found / parse error                  */
#define YH_E_SELFTEST -5    /* a Monte-Carlo self-test left its tolerance    */

/* ------------------------------------------------------------------------ */
/* Scene description (geometry level, no acceleration data).                 */
/* Mirrors the subset of yocto::pathtrace scene structs reachable from the   */
/* hair path (yocto_pathtrace.h:272-387).                                    */
/* ------------------------------------------------------------------------ */

/* ptr::shape (yocto_pathtrace.h:335-366). Exactly one of lines / triangles
 * is non-empty. For line shapes `normals` holds the hair TANGENTS
 * (yocto_pbrt.h:1782-1787) and `radius` must be given.                      */
typedef struct yh_shape {
  int          num_vertices;
  const float* positions; /* 3 * num_vertices                              */
  const float* normals;   /* 3 * num_vertices, or NULL                     */
  const float* radius;    /* num_vertices, or NULL (triangles)             */
  int          num_lines;
  const int*   lines;     /* 2 * num_lines                                 */
  int          num_triangles;
  const int*   triangles; /* 3 * num_triangles                             */
  const float* texcoords; /* 2 * num_vertices, or NULL (texcoord = element uv,
                             yocto_pathtrace.cpp:295-311)                   */
} yh_shape;

/* A colour texture (ptr::texture colorf / colorb, yocto_pathtrace.h:282-287):
 * RGB, row-major, top row first. Byte textures are sRGB-encoded unless the
 * lookup asks for linear (lookup_texture, yocto_pathtrace.cpp:147-164).      */
typedef struct yh_texture {
  int         width, height;
  int         is_byte; /* 1: pixels = uint8 RGB (colorb); 0: float RGB (colorf) */
  const void* pixels;
} yh_texture;

/* ptr::material (yocto_pathtrace.h:293-329), restricted to the lobes the hair
 * configs reach: emission, diffuse colour and the hair parameters
 * (yocto_extension.h:86-95). Specular, metallic, transmission / refraction,
 * delta (roughness 0) and opacity lobes follow yocto_pathtrace.cpp:405-471,
 * homogeneous volumes :498-533,1403-1414,1458-1497. Of the material textures
 * the colour ones are represented (emission_tex, color_tex, scattering_tex:
 * 1-based index into yh_scene_desc::textures, 0 = none); the scalar ones
 * (specular / metallic / roughness / opacity / normal maps) are not.         */
typedef struct yh_material {
  float emission[3];
  float color[3];
  float specular, metallic, roughness, transmission, opacity, ior;
  int   thin;
  float sigma_a[3];
  float beta_m, beta_n, alpha, eta, eumelanin, pheomelanin;
  /* homogeneous volume inside a closed surface (thin = 0 and transmission > 0,
   * yocto_pathtrace.cpp:498-533): density = -log(clamp(color, 1e-4, 1)) / trdepth */
  float scattering[3];
  float scanisotropy;
  float trdepth;      /* 0.01 (yocto_pathtrace.h:305)                          */
  int   emission_tex, color_tex, scattering_tex; /* 1-based, 0 = none          */
} yh_material;

/* ptr::object (yocto_pathtrace.h:369-373) */
typedef struct yh_object {
  float frame[12];
  int   shape;
  int   material;
} yh_object;

/* ptr::environment (yocto_pathtrace.h:376-380); texels = linear float RGB,
 * row-major, top row first, or NULL for a constant environment.             */
typedef struct yh_environment {
  float        frame[12];
  float        emission[3];
  int          tex_width, tex_height;
  const float* texels;
} yh_environment;

/* ptr::camera (yocto_pathtrace.h:272-278) */
typedef struct yh_camera {
  float frame[12];
  float lens;
  float film[2];
  float focus;
  float aperture;
} yh_camera;

typedef struct yh_scene_desc {
  int                   num_shapes;
  const yh_shape*       shapes;
  int                   num_materials;
  const yh_material*    materials;
  int                   num_objects;
  const yh_object*      objects;      /* in reference order (alphabetical)  */
  int                   num_environments;
  const yh_environment* environments;
  yh_camera             camera;
  int                   num_textures; /* material textures (environments carry their own) */
  const yh_texture*     textures;
} yh_scene_desc;

/* shader_type (yocto_pathtrace.h:177-182), same values as the reference's enum */
enum {
  YH_SHADER_NAIVE    = 0, /* trace_naive    (pt.cpp:1514-1581): brdf sampling only, no MIS, no volumes */
  YH_SHADER_PATH     = 1, /* trace_path     (pt.cpp:1380-1511): the hot path                            */
  YH_SHADER_EYELIGHT = 2, /* trace_eyelight (pt.cpp:1584-1641): light at the eye, delta chains only     */
  YH_SHADER_NORMAL   = 3, /* trace_normal   (pt.cpp:1644-1658): shading normal as a colour              */
  YH_SHADER_COUNT    = 4
};

/* trace_params (yocto_pathtrace.h:188-197). NOTE: a zeroed struct selects
 * shader 0 = naive, as a zeroed reference trace_params would; fill `shader`. */
typedef struct yh_trace_params {
  int      resolution; /* 720                                               */
  int      bounces;    /* 8                                                 */
  float    clamp;      /* 100                                               */
  uint64_t seed;       /* 961748941                                         */
  int      shader;     /* YH_SHADER_PATH; others: "sampler unknown" error   */
  int      hair_exact; /* extension (0 = default). The hair BSDF is evaluated within 1e-4 relative of the
                        * reference with hardware reciprocal / sqrt / log2 / sin / cos and float asinf; 1 selects
                        * the exact forms (IEEE divisions, library log / sin / cos, the reference's DOUBLE asin,
                        * yocto_extension.cpp:111,148-151): paths follow the reference's for more bounces at about
                        * 0.88 x the speed. Path shader only; runs the 512 x 4 quad kernel (csrc/exact.hip).        */
} yh_trace_params;

/* Per-sample work counters of the reference algorithm: what
 * the roofline's algorithmic-bytes figure is built from.                    */
typedef struct yh_workcounts {
  uint64_t samples;
  uint64_t rays;       /* scene-level intersect calls                       */
  uint64_t nodes;      /* bvh_node visits: scene + shape + instance BVHs    */
  uint64_t seg_tests;  /* intersect_line calls                              */
  uint64_t tri_tests;  /* intersect_triangle calls                          */
  uint64_t hair_shades;
  uint64_t surf_shades;
  uint64_t env_lookups;
  uint64_t env_samples;
  /* wave-level profile of the instrumented launch (diagnostics): shader-clock
   * cycles in traversal / shading summed over waves, 100 MHz ticks spent on
   * tiles, regeneration-loop iterations, traversal trip counts per wave (max
   * over lanes) and per lane (sum), live lanes per iteration                 */
  uint64_t cyc_trace, cyc_shade, ticks_tile, wave_iters, wave_steps, lane_steps, lane_iters;
  uint64_t cyc_geom, cyc_sample, cyc_eval, cyc_rest; /* split of cyc_shade */
  /* divergence of the traversal loop: (wave trips that ran the code, lanes
   * active in them) for wide-node, line-leaf, triangle-leaf, ENTER and
   * scene-node steps                                                         */
  uint64_t branch[10];
} yh_workcounts;

typedef struct yh_context yh_context;

/* ------------------------------------------------------------------------ */
/* Context                                                                    */
/* ------------------------------------------------------------------------ */

/* Creates a context on HIP device `device`. Returns NULL when no usable GPU
 * or the HIP code object is missing: there is NO CPU fallback.
 * Every call that waits for the device waits at most YHAIR_LAUNCH_TIMEOUT_S seconds (environment, default 1800): a launch that
 * does not complete in time returns YH_E_DEVICE, and from then on the context refuses every call that would touch the device
 * (a hung kernel cannot be recalled); yh_destroy of such a context returns at once and frees nothing. A caller that wants to
 * retry starts a fresh process. (The reference's trace_samples cannot hang: CPU threads over rows, yocto_pathtrace.cpp:1954-1989.) */
yh_context* yh_create(int device);
void        yh_destroy(yh_context* ctx);
/* Text of the last error on this context (or of the failed yh_create when
 * ctx == NULL). Never NULL.                                                 */
const char* yh_last_error(const yh_context* ctx);
/* Library version string and the code-object architecture it was built for. */
const char* yh_version(void);

/* ------------------------------------------------------------------------ */
/* Whole-path API: replaces yocto::pathtrace                                  */
/* ------------------------------------------------------------------------ */

/* init_bvh + init_lights (yocto_pathtrace.cpp:755-818, 1695-1740): builds the
 * two-level BVH (reference-identical binary middle-split tree, so that
 * closest-hit results, including exact-t ties, match the reference), the
 * area-light triangle CDFs and the environment texel CDF, precomputes inverse
 * object frames and per-material hair constants, and uploads everything. Shapes of 32 768 primitives and more are built ON THE
 * DEVICE from their vertex arrays as they are (bounds, tree, leaf-ordered records), and every shape's tree is collapsed there into
 * the 4- / 8- / 16-wide nodes the kernels traverse (csrc/bvh_gpu.hip): the call returns with everything a launch will read in
 * place (1.6 M hair segments: 41 ms); the host arrays are borrowed for the call only.
 * LIMITS (YH_E_INVALID with a message beyond them): a shape holds fewer than 2^27 elements; the traversal kernels address a
 * scene's trees as one array of 32-byte units with 27-bit leaf references and 30-bit node references — a line segment takes one
 * unit, a triangle two, a 4- / 8- / 16-wide node four / eight / sixteen — i.e. about 134 M segments or 67 M triangles in ALL shapes together (instances share
 * their shape); at most 4 environments and 16 lights. Scenes of more than ~46 objects or 24 materials run the GENERAL
 * kernel variants (their tables do not fit the LDS budget): slower, same pixels. The one-lane kernels (the streaming integrator
 * of dense hair, large closest-hit batches) address that array with 32-bit byte offsets: beyond 4 GB of it (about fifty million
 * segments) they are not candidates and the quad kernels render — same pixels.                                                */
int yh_upload_scene(yh_context* ctx, const yh_scene_desc* scene);

/* init_state (yocto_pathtrace.cpp:1931-1946): image size from the camera film
 * and params->resolution, zeroed accumulators, per-pixel PCG32 streams
 * make_rng(seed, rand1i(master, 1<<31)/2+1) with master = make_rng(1301081). */
/* On an image this context has not rendered yet it also launches a 1-sample
 * PROBE of the path shader (the per-tile costs that plan the first real launch)
 * and puts accumulators and RNG streams back as they were: blocking, one
 * kernel launch, no effect on the pixels.                                     */
int yh_init_state(yh_context* ctx, const yh_trace_params* params);
int yh_image_size(const yh_context* ctx, int* width, int* height);

/* Tile sharding for one-process-per-GPU rendering: this
 * context renders only the 8x8-pixel tiles with tile_id % world == rank.
 * Pixel results do not depend on (rank, world). Call before yh_init_state.   */
int yh_set_shard(yh_context* ctx, int rank, int world);

/* trace_samples called `nsamples` times (yocto_pathtrace.cpp:1992-2007, call
 * site apps/yscenetrace/yscenetrace.cpp:256-258): adds nsamples samples to
 * every owned pixel. Blocking. Normally one kernel launch; a request of 64
 * samples or more on an image whose kernels have not been timed yet starts
 * with 32-sample launches of the candidates (same samples, same bits:
 * yh_last_trace_ms reports the sum and the number of launches): up to three
 * kernels, each tried once, twice when two of them tie within 15 %.          */
int yh_trace_samples(yh_context* ctx, int nsamples);
/* Same, but only enqueues the work on the context's stream.                  */
int yh_trace_samples_async(yh_context* ctx, int nsamples);
int yh_synchronize(yh_context* ctx);

/* state->render (yocto_pathtrace.h:426-429): accumulated / samples, float4
 * per pixel, row-major top row first. Non-owned pixels are 0.                */
int yh_download(yh_context* ctx, float* rgba);
/* Packs the owned tiles' float4 pixels into a DEVICE buffer (the payload of
 * the RCCL gather). `capacity` in float4 pixels; *count receives the number
 * written. Tiles are in increasing tile_id order, 64 pixels per tile. This
 * call and yh_unpack_tiles_device run on the context's own non-blocking
 * stream and return when done: earlier writes to the buffers from other
 * streams (e.g. a torch allocation's fill) must have completed before.       */
int yh_pack_tiles_device(yh_context* ctx, void* device_rgba, int64_t capacity,
    int64_t* count);
/* Inverse on the gathering rank: scatters rank `src_rank`'s packed tiles into
 * a full W*H float4 DEVICE image.                                            */
int yh_unpack_tiles_device(yh_context* ctx, const void* device_packed,
    int src_rank, int world, void* device_image);
/* Number of float4 pixels yh_pack_tiles_device writes for (rank, world).     */
int64_t yh_shard_pixels(const yh_context* ctx, int rank, int world);

/* Multi-GPU inside ONE process (yscenetrace --gpus N; the reference renders on one device and has
 * no counterpart, the seam is save_image's input, apps/yscenetrace/yscenetrace.cpp:270): `n`
 * contexts, context i holding shard (i, n) of the same image (yh_set_shard) and the same number of
 * samples. Every context packs its tiles; ONE ncclGather over RCCL / xGMI (grouped, one call per
 * communicator; librccl is opened on first use) brings them to contexts[0], which un-interleaves
 * them and copies the full W*H float4 image to `rgba`. Contexts that share a device (tests on a
 * one-GPU box) or YHAIR_GATHER=peer use device-to-device copies instead of the collective.        */
int yh_gather_framebuffer(yh_context** contexts, int n, float* rgba);

/* Per-pixel state (yocto_pathtrace.h:419-423) for checkpoint / parity tests:
 * rng state words (2 x u64 per pixel) and sample count.                      */
int yh_download_rng(yh_context* ctx, uint64_t* state_inc);

/* Work counters of the launches since the last reset (instrumented build of
 * the same kernel; 0 = ok).                                                  */
int yh_trace_samples_counted(yh_context* ctx, int nsamples, yh_workcounts* out);

/* HIP-event time in milliseconds of the most recent yh_trace_samples launch
 * sequence on the context's own stream, and the number of kernel launches.   */
int yh_last_trace_ms(const yh_context* ctx, float* ms, int* launches);

/* Which sample-loop kernel the most recent yh_trace_samples launch ran (the host picks per launch from
 * measured times; every choice renders the same bits): 0 = k_trace, a quad of lanes per path, 512 threads
 * x 4 waves per SIMD; 1 = the same at 256 x 5 (dense images); 3 = k_stream, one lane per path (dense
 * images); 4 = k_trace with an OCTET per path over 8-wide BVH nodes, 7 = the same with leaf pairs, 6 =
 * SIXTEEN lanes per path over 16-wide nodes, 8 = the same with leaf groups (launches bound by the chain of
 * one path: few expensive pixels per GPU); 5 = side by side in one launch: the few items that top every
 * launch of a sparse image as octets, everything else as quads; 2 = quads over 8-wide nodes: a developer
 * build, never chosen (YHAIR_SHAPE=n forces a shape; YHAIR_DEVICE_SHARE=k tells the choice that k processes render on
 * this device at once). < 0 = nothing launched yet (or an error code). With
 * yh_trace_params::hair_exact it is always 0.                                                            */
int yh_launch_shape(const yh_context* ctx);
/* The measurements behind that choice on the current image: for launch shape k < count, the milliseconds per sample
 * of its fastest 32-sample trial launch (0 = not tried, < 0 = cannot run on this device) and the number of trials.
 * Returns the number of launch shapes (9), or a negative error code.                                               */
int yh_kernel_trials(const yh_context* ctx, double* ms_per_sample, int* trials, int count);
/* 1 while a candidate kernel of the current image still wants a timing trial — the next yh_trace_samples of 64 samples or
 * more will start with a 32-sample launch of it — else 0: a caller that times its launches (bench.py) keeps warming up
 * until this is 0. The record of an image is kept per process and, when the caller opted in (yh_set_trial_cache_dir), on disk.
 * Replaces nothing in the reference (host/launch_plan.cpp: pick_launch_shape).                                          */
int yh_trials_pending(const yh_context* ctx);
/* The kernel-trial record ON DISK (process-wide, opt-in: a library call writes no file unless asked to). `dir` = a directory
 * (created when needed) that holds trials_v2.txt, one appended line per image, keyed by device, the loaded library's
 * fingerprint, scene, image size, shard and bounces: an image found there runs no trial at all, so two processes (two
 * ranks, two runs) render one image with one kernel. NULL or "" = no file (the default). yscenetrace / ysceneitraces /
 * bench.py pass yh_default_trial_cache_dir() = $XDG_CACHE_HOME/yhair or ~/.cache/yhair. The environment's YHAIR_CACHE_DIR
 * names a directory too (and wins); YHAIR_NO_DISK_CACHE switches the file off whatever was set. No reference counterpart. */
int         yh_set_trial_cache_dir(const char* dir);
const char* yh_default_trial_cache_dir(void);

/* Load-balance telemetry: for every tile id (row-major over ceil(W/8) x
 * ceil(H/8) tiles) the time its wavefront spent on it in the most recent
 * launch (0 for tiles of other shards). The UNIT depends on the kernel that
 * ran (yh_launch_shape): k_trace (0, 1) reports ticks of the 100 MHz device
 * wall clock, k_stream (3) the BVH steps of the tile's rays — both are
 * relative costs for scheduling, comparable within one launch only.
 * `count` = number of tiles the caller's buffer holds.                       */
int yh_tile_costs(yh_context* ctx, uint32_t* ticks, int count);
/* The same per WORK ITEM (a tile's four 4x4-pixel quadrants, item = 4 * tile + quadrant: what a wavefront takes
 * at a time and what the launch's hand-out order is planned from). `count` >= 4 * number of tiles.              */
int yh_item_costs(yh_context* ctx, uint32_t* costs, int count);

/* ------------------------------------------------------------------------ */
/* Unit-level API: replaces yocto::extension and the intersect_* functions    */
/* (batched, host arrays in / host arrays out; device does the arithmetic)    */
/* ------------------------------------------------------------------------ */

/* hair_brdf as 30 floats: sigma_a[3] alpha eta h v[4] s sin_2k_alpha[3]
 * cos_2k_alpha[3] gamma_o world_to_brdf[12]  (yocto_extension.h:97-113)      */
#define YH_HAIR_BRDF_FLOATS 30

/* The surface lobes of yocto_math.h:1513-1620 (implementation 4307-4755).    */
enum {
  YH_LOBE_DIFFUSE            = 0, /* eval/sample/_pdf  diffuse_reflection         */
  YH_LOBE_SPECULAR           = 1, /* microfacet_reflection(ior, ...)              */
  YH_LOBE_METAL              = 2, /* microfacet_reflection(eta, etak, ...)        */
  YH_LOBE_TRANSMISSION       = 3, /* microfacet_transmission                      */
  YH_LOBE_REFRACTION         = 4, /* microfacet_refraction                        */
  YH_LOBE_DELTA_SPECULAR     = 5, /* delta_reflection(ior, ...)                   */
  YH_LOBE_DELTA_METAL        = 6, /* delta_reflection(eta, etak, ...)             */
  YH_LOBE_DELTA_TRANSMISSION = 7, /* delta_transmission                           */
  YH_LOBE_DELTA_REFRACTION   = 8, /* delta_refraction                             */
  YH_LOBE_COUNT              = 9
};
/* yh_surface_bsdf_batch output per item: diffuse[3] specular[3] metal[3]
 * transmission[3] refraction[3] roughness opacity, the five lobe pdfs, then
 * f*|cos| [3], pdf and the sampled incoming [3] (delta forms when roughness
 * is 0, pt.cpp:495).                                                        */
#define YH_SURFACE_BSDF_FLOATS 29

/* eval_hair_brdf (yocto_extension.cpp:127-177). materials: n x yh_material
 * (only hair fields read); v: n; normal, tangent: 3n; out: 30n.              */
int yh_hair_brdf_batch(yh_context* ctx, int n, const yh_material* materials,
    const float* v, const float* normal, const float* tangent, float* brdf);
/* eval_hair_scattering (yocto_extension.cpp:255-336): out 3n.                */
int yh_hair_eval_batch(yh_context* ctx, int n, const float* brdf,
    const float* outgoing, const float* incoming, float* f);
/* sample_hair_scattering (yocto_extension.cpp:399-479): rn 2n -> incoming 3n */
int yh_hair_sample_batch(yh_context* ctx, int n, const float* brdf,
    const float* outgoing, const float* rn, float* incoming);
/* sample_hair_scattering_pdf (yocto_extension.cpp:481-551): out n.           */
int yh_hair_pdf_batch(yh_context* ctx, int n, const float* brdf,
    const float* outgoing, const float* incoming, float* pdf);
/* README.md:20 / BASELINE.json call it eval_hair_scattering_pdf: same entry. */
int yh_hair_eval_pdf_batch(yh_context* ctx, int n, const float* brdf,
    const float* outgoing, const float* incoming, float* pdf);

/* The pbrt `curve` -> hair-line conversion of the reference's pbrt loader
 * (libs/yocto/yocto_pbrt.h:1751-1797): each curve's first four control points
 * become a strand of five vertices (Bezier at u = 0, 1/4, 1/2, 3/4, 1), with
 * tangents as "normals" and radius = lerp(width0, width1, u), joined by four
 * lines. P: 12n; width0, width1: n; out: positions 15n, normals 15n, radius
 * 5n, lines 8n ints (vertex indices start at base_vertex + 5 * curve).       */
int yh_curves_to_lines(yh_context* ctx, int n, const float* P, const float* width0,
    const float* width1, int base_vertex, float* positions, float* normals,
    float* radius, int* lines);

/* build_bvh (yocto_pathtrace.cpp:598-650) on the host, exactly as
 * yh_upload_scene builds it: boxes = n x (min[3], max[3]). Call with nodes =
 * NULL to get the node count; nodes = 8 floats per node (bbox min, bbox max,
 * then as int bits: start, num | internal << 16 | axis << 24), primitives = n
 * ints (leaf order). Needs no GPU and no context. Returns the node count.     */
int yh_bvh_build(int n, const float* boxes, float* nodes, int* primitives);
/* The same tree as the device traverses it: `width` (4, 8 or 16) children per node = two, three or four levels of
 * the binary tree collapsed into one record of `width` 32-byte slots {min.xyz, max.x} {max.yz, ref, axes}
 * (yocto-hair_amd/host/bvh_build.h). ref: 0xFFFFFFFF empty; top two bits set = leaf (count << 27 | first
 * primitive position); else the index of the child node. axes: the split axes of the collapsed binary nodes, from
 * which a traversal ranks the children into the reference's near-first order (yocto_pathtrace.cpp:887-893).
 * Writes width * 8 floats per node to `slots` (NULL: count only); returns the number of nodes. No GPU needed.       */
int yh_bvh_build_wide(int n, const float* boxes, int width, float* slots);

/* The same tree built on the GPU (csrc/bvh_gpu.hip; what yh_upload_scene uses for
 * shapes of 32 768 primitives and more). Same arguments and result as yh_bvh_build;
 * the two are compared node for node in the tests.                            */
int yh_bvh_build_gpu(yh_context* ctx, int n, const float* boxes, float* nodes, int* primitives);
/* ... and its wide collapse made on the GPU too (csrc/bvh_gpu.hip: what yh_upload_scene runs for EVERY shape since round 6 — the host's
 * collapse_wide* of yh_bvh_build_wide are the restatement it is tested against). Same arguments and result as yh_bvh_build_wide, in the form the
 * traversal kernels read: a child's ref is the index of its FIRST SLOT (width x the child node's index), and for width 4 bits 8-11 of `axes`
 * hold the occupied slots.                                                                                                              */
int yh_bvh_build_wide_gpu(yh_context* ctx, int n, const float* boxes, int width, float* slots);

/* One surface lobe (kind = YH_LOBE_*) of yocto_math.h:1513-1620 (implementation
 * 4427-4755): eval_* (value times |cos|), sample_*_pdf and sample_* in one
 * call. params: 8n (ior, roughness [= brdf.roughness, already squared], eta[3],
 * etak[3]); normal, outgoing, incoming: 3n; rn: 3n (rnl, rn.x, rn.y);
 * out: 7n (f[3], pdf, sampled incoming[3]).                                   */
int yh_surface_lobe_batch(yh_context* ctx, int kind, int n, const float* params,
    const float* normal, const float* outgoing, const float* incoming,
    const float* rn, float* out);
/* The lobe mixture of a non-hair material: eval_brdf (yocto_pathtrace.cpp:
 * 405-471) followed by eval_brdfcos / sample_brdfcos / sample_brdfcos_pdf or,
 * for a delta mixture, eval_delta / sample_delta / sample_delta_pdf
 * (:1069-1280). out: YH_SURFACE_BSDF_FLOATS per item.                          */
int yh_surface_bsdf_batch(yh_context* ctx, int n, const yh_material* materials,
    const float* normal, const float* outgoing, const float* incoming,
    const float* rn, float* out);
/* intersect_scene_bvh (yocto_pathtrace.cpp:934-1046) on the uploaded scene.
 * rays: 8n floats (o[3] d[3] tmin tmax). Outputs per ray: object, element
 * (-1 on miss), uv[2], distance.                                             */
int yh_intersect_batch(yh_context* ctx, int n, const float* rays, int* object,
    int* element, float* uv, float* distance);

/* The four Monte-Carlo self-tests of yocto_extension.cpp:555-693 on the
 * device: 0 white_furnace, 1 white_furnace_sampled, 2 sampling_weights,
 * 3 sampling_consistency. Same seeds, counts and thresholds. `worst` (may be
 * NULL) receives the statistic furthest from its target. Returns YH_OK or
 * YH_E_SELFTEST ("TEST FAILED!").                                            */
int yh_selftest(yh_context* ctx, int which, float* worst);

/* ------------------------------------------------------------------------ */
/* Host-side scene I/O (C++ host code, no device work): the minimal JSON +    */
/* PLY + Radiance-HDR reader for the hair scenes, with the reference loader's */
/* semantics (yocto_sceneio.cpp:1064-1418: alphabetical objects, lookat,      */
/* add_radius 0.001, quads_to_triangles).                                     */
/* ------------------------------------------------------------------------ */
typedef struct yh_scene_file yh_scene_file;
yh_scene_file*       yh_scene_load(const char* json_path, const char* camera,
          char* error, int error_len);
const yh_scene_desc* yh_scene_get(const yh_scene_file* scene);
void                 yh_scene_free(yh_scene_file* scene);
/* save_image for .pfm (3 channels, top row first as the reference writes it,
 * yocto_image.cpp:1527-1556) and .hdr.                                       */
int yh_save_image(const char* path, int width, int height, const float* rgba,
    char* error, int error_len);

#ifdef __cplusplus
}
#endif
#endif /* YHAIR_H_ */
