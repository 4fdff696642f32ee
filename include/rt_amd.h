/*
 * rt_amd.h - C-ABI of the MI355X-native per-pixel hot path.
 *
 * The reference (DanielBarLev2/Ray-Tracing) has no FFI; the "interface for this
 * path" is the Python call sequence of main() (ray_tracer.py:20-33) and the
 * object model it passes around.  Each entry point below names the reference
 * call it replaces.  Plain C only: pointers + sizes, no C++/torch types, every
 * function returns 0 on success or a negative rt_status; the message for the
 * last failure on a context is available through rt_last_error().
 *
 * All reference arithmetic is float64 and ids are the reference's
 * `surface.index`; the ABI keeps both (double / int32).
 *
 * The same rt_scene_desc is consumed by the CPU oracle (oracle/oracle.cpp,
 * test infrastructure only) so that both sides see identical bytes.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_AMD_ABI_VERSION 2

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,  /* null pointer, negative count, inconsistent scene */
    RT_ERR_NO_DEVICE = -2,    /* no HIP device / wrong ordinal */
    RT_ERR_HIP = -3,          /* a HIP runtime call failed (text in rt_last_error) */
    RT_ERR_OOM = -4,          /* device allocation failed */
    RT_ERR_QUEUE_OVERFLOW = -5, /* secondary-ray queues exceeded RT_MAX growth; nothing written */
    RT_ERR_UNSUPPORTED = -6
} rt_status;

/* surf_kind values: what sits at list position k of the reference's `surfaces`
 * list (Parser.py:29,52-63).  Normals and materials are looked up by POSITION
 * using the hit's id (SurfaceAbs.py:61,83 - SURVEY §8a Q8). */
enum { RT_SURF_BACKGROUND = 0, RT_SURF_SPHERE = 1, RT_SURF_BOX = 2, RT_SURF_PLANE = 3 };

/*
 * Flattened scene: what Parser.parse_scene_file (Parser.py:25-71) returns plus
 * the tree of BSPNode.build_bsp_tree (BSPNode.py:54-121), as SoA float64.
 * The caller keeps ownership; rt_create copies everything.
 */
typedef struct rt_scene_desc {
    int32_t abi_version;         /* RT_AMD_ABI_VERSION */

    /* surfaces list by position (index 0 is Background in parsed scenes) */
    int32_t n_surfaces;
    const int32_t *surf_kind;    /* [n_surfaces] RT_SURF_* */
    const int32_t *surf_geom;    /* [n_surfaces] row in the sphere/box/plane table, -1 for background */
    const int32_t *surf_mat;     /* [n_surfaces] SurfaceAbs.get_material_index() (SurfaceAbs.py:38-42) */

    /* Sphere.py:6-8,135-139 */
    int32_t n_spheres;
    const double *sph_center;    /* [n_spheres*3] */
    const double *sph_radius;    /* [n_spheres]   */
    const double *sph_r2;        /* [n_spheres] radius ** 2 exactly as Python evaluates it (Sphere.py:79) */
    const int32_t *sph_id;       /* [n_spheres] surface.index */

    /* Cube.py:7-14,89-90 */
    int32_t n_boxes;
    const double *box_center;    /* [n_boxes*3] */
    const double *box_scale;     /* [n_boxes]   */
    const double *box_lo;        /* [n_boxes*3] position - scale/2 */
    const double *box_hi;        /* [n_boxes*3] position + scale/2 */
    const int32_t *box_id;       /* [n_boxes] */

    /* InfinitePlane.py:6-10: normal already normalised, d = -offset/|n|.
     * Order = order in the root node's inf_plains (BSPNode.py:111), Background excluded. */
    int32_t n_planes;
    const double *pln_normal;    /* [n_planes*3] */
    const double *pln_d;         /* [n_planes]   */
    const int32_t *pln_id;       /* [n_planes]   */
    int32_t far_clip;            /* 1 if a Background surface is in the root plane list (Background.py:23) */

    /* Material.py:4-11,23-62; list position == material index */
    int32_t n_materials;
    const double *mat_kd;        /* [n_materials*3] */
    const double *mat_ks;        /* [n_materials*3] */
    const double *mat_kr;        /* [n_materials*3] */
    const int32_t *mat_phong;    /* [n_materials] shininess truncated toward zero (Material.py:49,58) */
    const double *mat_transparency; /* [n_materials] */

    /* Light.py:7-14 */
    int32_t n_lights;
    const double *lgt_pos;       /* [n_lights*3] */
    const double *lgt_color;     /* [n_lights*3] */
    const double *lgt_spec;      /* [n_lights] specular_intensity */
    const double *lgt_shadow;    /* [n_lights] shadow_intensity   */
    const double *lgt_radius;    /* [n_lights] */

    /* BSP tree, DFS pre-order: left child of an internal node i is i+1.
     * Finite primitives are numbered spheres first, then boxes:
     * prim p < n_spheres is sphere p, else box p - n_spheres. */
    int32_t n_nodes;
    const double *node_lo;       /* [n_nodes*3] small_bound */
    const double *node_hi;       /* [n_nodes*3] big_bound   */
    const int32_t *node_right;   /* [n_nodes] index of the right child, -1 for a leaf */
    const int32_t *node_first;   /* [n_nodes] leaf: first entry in leaf_prim, else -1 */
    const int32_t *node_count;   /* [n_nodes] leaf: number of entries */
    int32_t n_leaf_refs;
    const int32_t *leaf_prim;    /* [n_leaf_refs] primitive numbers in leaf-list order */

    /* Camera.py:5-13 (basis from util.create_orthonormal_basis, util.py:48-75) */
    double cam_pos[3], cam_x[3], cam_y[3], cam_z[3];
    double cam_screen_distance, cam_screen_width;

    /* SceneSettings.py:1-5 */
    double background[3];
    int32_t sh_rays;             /* root_number_shadow_rays */
    int32_t rec_max;             /* max_recursions */
} rt_scene_desc;

#define RT_USE_SCENE_VALUE (-2147483647 - 1)

/* Per-frame parameters.  A frame is W*H pixels; with world > 1 this context
 * renders only the tiles t (row-major tile index) with t % world == rank and
 * returns them packed tile after tile (rt_tile_layout describes the order). */
typedef struct rt_render_params {
    int32_t width, height;
    int32_t sh_rays;     /* RT_USE_SCENE_VALUE or an override of root_number_shadow_rays */
    int32_t rec_max;     /* RT_USE_SCENE_VALUE or an override of max_recursions */
    int32_t jitter;      /* 0: deviations of Light.py:277-279 forced to zero (deterministic parity mode);
                            1: counter-based hash, 6 x U(-g,g) per shadow ray */
    uint32_t seed;
    int32_t rank, world; /* tile interleave over ranks; 0,1 = whole image */
    int32_t tile_w, tile_h; /* 0 = default 8x8 */
    int32_t flags;       /* RT_FLAG_* */
    int32_t reserved;
} rt_render_params;

#define RT_FLAG_EXACT_TRAVERSAL 1  /* diagnostic: visit nodes with the reference's float64 slab test only
                                      (no float32 culling, no pruning); results must be identical */
#define RT_FLAG_NO_BEAM 4          /* diagnostic: decide every shadow ray with its own float64-confirmed traversal
                                      instead of the per-(hit,light) beam + float32 filtered predicates */
#define RT_FLAG_NO_OVERLAP 8       /* diagnostic: run the recursion levels strictly one after the other (the default lets the
                                      deeper levels run beside the soft shadows of level 0); same result */
#define RT_FLAG_NO_SMALL 16         /* diagnostic: scenes small enough for the LDS-resident fused path (one kernel per level, used for
                                      s*s <= 4) take the general wavefront pipeline instead; same result */
#define RT_FLAG_TIME_DOMINANT 32   /* HIP events around the k_beam_setup launches only, schedule unchanged (levels stay overlapped):
                                      rt_stats.ms_beam_setup / beam_setup_launches of that frame (bench.py: live duration of the dominant kernel) */
#define RT_FLAG_NO_TILES 64        /* diagnostic: every (hit, light) pair of level 0 walks the tree itself instead of filtering the
                                      candidate row that the binning kernels made for its 8 x 8 pixel tile; same result */
#define RT_FLAG_PROFILE 2          /* bracket every kernel launch with HIP events on the launch stream and
                                      report per-family device time in rt_stats (used by bench.py's roofline) */

/* Work counters of the last frame (the Mrays/s numerator counts rays exactly as
 * the reference issues them to BSPNode.intersect_vectorize). */
typedef struct rt_stats {
    int64_t primary_rays, shadow_rays, reflect_rays, transmit_rays;
    int64_t hits;                /* shaded hits over all levels */
    int64_t levels;              /* recursion levels that had live rays */
    int64_t max_queue;           /* largest ray queue of the frame */
    double ms_total;             /* device time of the frame (HIP events) */
    double ms_intersect, ms_shadow, ms_shade; /* per kernel family, summed over levels (RT_FLAG_PROFILE only) */
    double ms_gen, ms_resolve;   /* RT_FLAG_PROFILE only */
    int64_t launches_per_family; /* RT_FLAG_PROFILE only: number of k_intersect (= k_shadow = k_shade) launches */
    double ms_beam_setup;        /* RT_FLAG_PROFILE / RT_FLAG_TIME_DOMINANT: HIP events around the k_beam_setup launches alone (the frame's dominant kernel) */
    int64_t beam_setup_launches; /* RT_FLAG_PROFILE / RT_FLAG_TIME_DOMINANT */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* Replaces: scene objects living in Python + BSPNode tree (ray_tracer.py:20-22). */
int rt_create(const rt_scene_desc *scene, int device, rt_ctx **out);
int rt_destroy(rt_ctx *ctx);
/* Message for the last failure on ctx (ctx == NULL: last rt_create failure of this thread). */
const char *rt_last_error(const rt_ctx *ctx);
int rt_abi_version(void);

/* Number of pixels this rank renders and, optionally, their global pixel index
 * (row*W + col) in output order.  Pure host arithmetic. */
int64_t rt_tile_pixels(const rt_render_params *p);
int rt_tile_layout(const rt_render_params *p, int64_t *pixel_index /* [rt_tile_pixels] */);

/*
 * Replaces ray_tracer.py:25-33: get_initial_rays (ray_functions.py:7-43) +
 * ray_tracing(...).clip(0,1) (ray_tracer.py:36-133) + save_image's quantisation
 * (util.py:94-98).  Host buffers, blocking.  Any output may be NULL.
 *   rgb_out  [npix*3] float32, clipped to [0,1]
 *   rgb8_out [npix*3] uint8 = (x*255) truncated
 *   ids_out  [npix]   int32 primary hit ids (0 = background)
 * npix = rt_tile_pixels(p); output order = rt_tile_layout (row-major when world == 1).
 */
int rt_render(rt_ctx *ctx, const rt_render_params *p, float *rgb_out, uint8_t *rgb8_out, int32_t *ids_out,
              rt_stats *stats);

/*
 * Mean of `frames` stochastic frames: the frames rt_render gives for seed, seed+1, ...,
 * seed+frames-1, each clipped to [0,1] (ray_tracer.py:30), summed in float64 on the device and divided by `frames`
 * before the float32 / uint8 outputs are written (util.py:98 for the quantisation).  The reference renders one frame
 * per run; this is the accumulation its soft shadows need to converge (and what the PSNR parity check averages).
 * stats: ray counters and ms_total summed over the frames.  Host buffers, blocking, either output may be NULL.
 */
int rt_render_mean(rt_ctx *ctx, const rt_render_params *p, int32_t frames, float *rgb_out, uint8_t *rgb8_out,
                   rt_stats *stats);

/* Same, outputs are DEVICE pointers on ctx's device (any may be NULL) and the
 * work is only enqueued on `stream` (a hipStream_t, NULL = the context's own
 * stream); nothing is synchronised.  Used by the bench and the multi-GPU gather.  Frames of one context share its
 * queues and framebuffers: enqueue them on ONE stream (they run back to back), use two contexts for two streams.
 * A frame enqueued behind a frame that is still pending may start its camera pass beside that frame's last kernels (on a
 * stream of the context, on a second copy of the level-0 state; frames without d_ids_out only): the outputs are written on
 * `stream`, in call order, and every frame equals the frame rendered alone ("frames in flight"). */
int rt_render_device(rt_ctx *ctx, const rt_render_params *p, float *d_rgb_out, uint8_t *d_rgb8_out,
                     int32_t *d_ids_out, void *stream);
/* Wait for the context's stream and fetch the counters of the last frame. */
int rt_sync_stats(rt_ctx *ctx, rt_stats *stats);

/* Frames voided by a queue overflow (RT_ERR_QUEUE_OVERFLOW) since the context was created or the counter was last reset,
 * counted on the device at the end of every frame - also of frames whose counters nobody fetched with rt_sync_stats (a
 * timing loop checks it ONCE after its last frame: a voided frame returns early and would be timed as a fast one).
 * Synchronises the device. */
int rt_voided_frames(rt_ctx *ctx, int reset, int64_t *count);

/* Device bytes of the frame state (ray queues, hit and pair records, framebuffers) the context currently holds. */
int64_t rt_frame_bytes(const rt_ctx *ctx);

/*
 * rt_render_device for callers that cannot allocate.  No counterpart in the reference: it exists for code under HIP graph
 * capture, tight serving loops and harnesses that forbid hipMalloc.  Same arguments, same image (the levels run in sequence
 * on one buffer set; frames are bit-identical to rt_render_device's), but every buffer the frame's kernels need beyond the
 * caller's outputs - ray queues, hit records, pair records, lit counters, framebuffers, the counter block - is carved out of
 * one static device array inside the library (rt_noalloc_scratch_bytes() = 64 MiB per GPU, held until the process ends)
 * instead of hipMalloc'ed memory.  The call issues no hipMalloc / hipFree / event or stream creation.
 *   limit: the carve for (pixel slots, shadow-casting lights) must fit the array - about 4.4 KB per pixel slot with 5 such
 *     lights (~15 000 slots, e.g. 128 x 96), 1.4 KB with one (~47 000 slots); larger frames -> RT_ERR_UNSUPPORTED with a
 *     message naming the frame.  RT_FLAG_PROFILE -> RT_ERR_UNSUPPORTED.
 *   queues hold 2 x the pixel slots and cannot grow: RT_ERR_QUEUE_OVERFLOW from rt_sync_stats is final for that frame.
 *   every call on a GPU uses that GPU's one array (whatever the context): calls that overlap on one GPU - two streams, two
 *     contexts, two threads - race; run them back to back on one stream or synchronise between them.
 *   what the call enqueues besides kernels and memsets: the context's two timing events and one copy of the counter block to
 *     pinned host memory, for rt_sync_stats.  rt_sync_stats describes a call that EXECUTED on its stream; for a call recorded
 *     into a HIP graph the counters are those of the last replay that finished, ms_total is not available, and the library
 *     has no test under capture (allocation-free is what is tested: rt_debug_alloc_calls unchanged, tests/test_zz_no_alloc.py).
 */
int rt_render_device_noalloc(rt_ctx *ctx, const rt_render_params *p, float *d_rgb_out, uint8_t *d_rgb8_out,
                             int32_t *d_ids_out, void *stream);
int64_t rt_noalloc_scratch_bytes(void);
/* Diagnostic: number of allocating HIP calls (hipMalloc, hipHostMalloc, event creation) this library has issued in the process. */
int64_t rt_debug_alloc_calls(void);
/* Diagnostic: frames of this context that took the tile beams (candidates per 8 x 8 pixel tile by binning, see RT_FLAG_NO_TILES),
 * counted when their counters are fetched (rt_sync_stats) and no queue overflow voided them. */
int64_t rt_debug_tile_frames(const rt_ctx *ctx);
/* Diagnostic: frames of this context whose camera pass was enqueued on the library's second stream, on the other copy of the
 * level-0 state, so that it may run beside the tail of the frame before (frames back to back on one stream; rt_render_device). */
int64_t rt_debug_early_frames(const rt_ctx *ctx);

/*
 * Replaces ray_tracing(rays_sources, rays_directions, ...) (ray_tracer.py:36-133)
 * for an arbitrary ray batch: rgb_out[n*3] float64, unclipped.  Ray i plays the
 * role of pixel i for the jitter hash.
 */
int rt_trace(rt_ctx *ctx, const rt_render_params *p, const double *origins, const double *directions, int64_t n,
             double *rgb_out, rt_stats *stats);

/*
 * Replaces BSPNode.intersect_vectorize (BSPNode.py:201-210): nearest hit of n
 * rays.  points_out[n*3] float64 (the reference's miss point (1e8,1e8,1e8) when
 * far_clip, NaN otherwise), ids_out[n] int64 (0 = miss).
 */
int rt_intersect(rt_ctx *ctx, const double *origins, const double *directions, int64_t n, double *points_out,
                 int64_t *ids_out);

/*
 * The any-hit side of the same operator: the visibility test Light.compute_shadows applies to
 * every shadow ray (Light.py:214-221).  lit_out[i] = 1 iff the nearest hit X of ray i = (origins[i], directions[i])
 * (BSPNode.intersect_vectorize) satisfies ||X - targets[i]|| < 1e-6, else 0 (NaN anywhere -> 0).  Same verdicts as
 * rt_intersect followed by that comparison on the host; the walk ends at the first float64-confirmed hit in front
 * of the target and skips everything beyond it.  Rays as Light.get_shadow_rays makes them
 * (direction = (target - origin) / (||target - origin|| + 1e-6)) are the fast case; any other batch is served too.
 */
int rt_visibility(rt_ctx *ctx, const double *origins, const double *directions, const double *targets, int64_t n,
                  uint8_t *lit_out);

/*
 * The two batched queries on DEVICE buffers (PyTorch / HIP callers): same kernels, same results as
 * rt_intersect / rt_visibility, but nothing crosses PCIe and nothing is allocated - the kernel is only enqueued on
 * `stream` (a hipStream_t, NULL = the context's own stream) and nothing is synchronised.  Buffers: C-contiguous float64
 * [n*3] on ctx's device (ids int64 [n], lit uint8 [n]).  Replaces BSPNode.intersect_vectorize (BSPNode.py:201-210) and the
 * lit test of Light.compute_shadows (Light.py:214-221) for callers whose rays already live in HBM.
 */
int rt_intersect_device(rt_ctx *ctx, const double *d_origins, const double *d_directions, int64_t n, double *d_points_out,
                        int64_t *d_ids_out, void *stream);
int rt_visibility_device(rt_ctx *ctx, const double *d_origins, const double *d_directions, const double *d_targets, int64_t n,
                         uint8_t *d_lit_out, void *stream);

/* Rank-0 side of the multi-GPU gather: scatter `world` packed per-rank tile
 * buffers (device memory, rank r at d_packed + r*stride_bytes) into a row-major
 * W*H*3 uint8 image on the device. */
int rt_deinterleave_u8(rt_ctx *ctx, const rt_render_params *p, const uint8_t *d_packed, int64_t stride_bytes,
                       uint8_t *d_image, void *stream);

/* Replaces get_initial_rays(camera, W, H) (ray_functions.py:7-43) as a standalone query:
 * sources_out / directions_out [W*H*3] float64, row-major pixel order (i = row, j = col). */
int rt_initial_rays(rt_ctx *ctx, int width, int height, double *sources_out, double *directions_out);

/*
 * Kernel-level entry points (no context): ONE primitive against a ray batch, through the same device
 * functions the frame kernels use.  params12 = 12 doubles:
 *   RT_SURF_SPHERE {cx,cy,cz, r**2, r, 0...}   Sphere.intersect_vectorized / calculate_normals (Sphere.py:63-133)
 *   RT_SURF_BOX    {lo xyz, hi xyz, centre xyz, scale, 0,0}   Cube.intersect_vectorized / calculate_normals
 *                                                             (Cube.py:78-93,177-207)
 *   RT_SURF_PLANE  {n xyz (unit), d, 0...}     InfinitePlane.intersect_vectorized (InfinitePlane.py:60-92)
 * points_out rows are NaN where the ray misses.
 */
int rt_unit_prim_intersect(int device, int kind, const double *params12, const double *origins,
                           const double *directions, int64_t n, double *points_out);
int rt_unit_prim_normals(int device, int kind, const double *params12, const double *points, int64_t n,
                         double *normals_out);
/* Cube.cube_intersection_mask (Cube.py:95-138): mask_out[n] (0/1), t_near_out[n] */
int rt_unit_slab_mask(int device, const double *lo, const double *hi, int outside_hits_only, const double *origins,
                      const double *directions, int64_t n, uint8_t *mask_out, double *t_near_out);

/*
 * Tree build on the GPU.  Replaces BSPNode.build_bsp_tree (BSPNode.py:99-121) and its
 * recursion build_bsp_tree_rec (BSPNode.py:54-97): the finite objects' enclosing corners
 * (Object3D.get_enclosing_values) in list order go in, the tree comes out in the DFS pre-order layout of
 * rt_scene_desc (node_lo/hi/right/first/count + leaf lists), bit-identical to the recursive host build:
 * same float64 cut locations, same membership, same order inside every leaf.
 *   small_corner / big_corner [n*3] float64 (host), root_lo / root_hi [3] or NULL (= min / max over the
 *   corners, BSPNode.py:113-118), leaf_size = 10 and epsilon = util.EPSILON in the reference (BSPNode.py:67).
 * leaf_prim holds POSITIONS into the caller's corner arrays.  The result stays in HBM until rt_bsp_fetch
 * copies it into host arrays sized from rt_bsp_stats (any pointer may be NULL).
 */
typedef struct rt_bsp rt_bsp;
typedef struct rt_bsp_stats {
    double ms_total;       /* host wall clock of rt_bsp_build: uploads, kernels, per-level read-backs */
    double ms_device;      /* HIP events around all kernels of the build on its stream */
    int32_t n_nodes, n_leaf_refs, depth, launches;
    int64_t ref_visits;    /* sum over the levels of the references classified there */
    int64_t bytes_scratch; /* HBM taken during the build (freed on return, except the result) */
} rt_bsp_stats;
int rt_bsp_build(int device, int64_t n, const double *small_corner, const double *big_corner, const double *root_lo,
                 const double *root_hi, int leaf_size, double epsilon, rt_bsp **out);
int rt_bsp_stats_get(const rt_bsp *tree, rt_bsp_stats *stats);
int rt_bsp_fetch(rt_bsp *tree, double *node_lo, double *node_hi, int32_t *node_right, int32_t *node_first,
                 int32_t *node_count, int32_t *leaf_prim);
int rt_bsp_destroy(rt_bsp *tree);
/* Message of the last failure on `tree` (NULL: last rt_bsp_build failure of this thread). */
const char *rt_bsp_last_error(const rt_bsp *tree);

/*
 * Scene ingestion without Python objects.  Replaces Parser.parse_scene_file
 * (Parser.py:25-71) plus the flattening into rt_scene_desc: the scene text goes in, the SoA tables come out,
 * bit-identical to what the Python path derives (same index rules: one counter for sph/pln/box and lgt lines
 * from 1, Background at list position 0, materials sorted by index with the `set` line's background material
 * first).  Errors (unknown line type, a token float() would refuse, too few numbers) name the line.
 *   threads <= 0: RT_AMD_PARSE_THREADS or min(16, hardware threads); the text is cut at line ends.
 * The tree is attached afterwards, either built on the GPU from the scene's own corner arrays
 * (rt_scene_build_tree = rt_bsp_build + rt_bsp_fetch + position -> primitive-number mapping) or handed in
 * (rt_scene_set_tree, leaf entries as positions in the finite-object order, as BSPNode keeps them).
 * rt_scene_get_desc stays valid until rt_scene_free; pass it to rt_create.
 */
typedef struct rt_scene rt_scene;
typedef struct rt_scene_info {
    int64_t lines;        /* text lines read */
    int64_t n_records;    /* non-empty, non-comment lines */
    int64_t n_finite;     /* spheres + boxes = entries of the corner arrays */
    int32_t threads, has_camera, has_settings, reserved;
} rt_scene_info;
int rt_scene_parse(const char *text, int64_t len, int threads, rt_scene **out);
int rt_scene_parse_file(const char *path, int threads, rt_scene **out);
const rt_scene_desc *rt_scene_get_desc(const rt_scene *scene);
int rt_scene_get_info(const rt_scene *scene, rt_scene_info *info);
/* Object3D.get_enclosing_values of the finite objects in list order (Sphere.py:135-139, Cube.py:209-213). */
int rt_scene_corners(const rt_scene *scene, const double **small_corner, const double **big_corner, int64_t *n);
/* Light.index per light and Material.index per material row (Parser.py:54-66). */
int rt_scene_light_ids(const rt_scene *scene, const int32_t **light_ids, const int32_t **material_ids);
int rt_scene_set_tree(rt_scene *scene, int32_t n_nodes, const double *node_lo, const double *node_hi,
                      const int32_t *node_right, const int32_t *node_first, const int32_t *node_count,
                      int32_t n_leaf_refs, const int32_t *leaf_positions);
int rt_scene_build_tree(rt_scene *scene, int device, rt_bsp_stats *stats /* may be NULL */);
void rt_scene_free(rt_scene *scene);
const char *rt_scene_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
