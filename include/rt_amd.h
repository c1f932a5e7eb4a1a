/*
 * rt_amd.h -- C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * The reference (piemot/raytracing) has no FFI; the seam this library replaces is
 *     pub fn render(&mut self, world: &impl Hittable, lights: Rc<dyn Hittable>)   src/camera.rs:368
 * i.e. the pixel/stratum loop of camera.rs:386-401 together with everything it calls
 * (get_ray :409, ray_color :432, Hittable::hit, Material::*, PDF::*).  A reference-side
 * binding flattens its scene graph into the arrays below, calls rt_render() once per
 * stratum range, and hands the returned film to its own ImageWriter (export.rs:5-8).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * negative code on failure (nothing unwinds across the ABI); rt_last_error() gives the text.
 * One handle per GPU.  Calls on one handle are not re-entrant; different handles may be
 * driven from different host threads.  All `const T*` arguments are caller-owned HOST
 * memory unless the name says device.
 *
 * Part 1 (rt_*)  : device backend -- scene upload, wavefront render, film, parity hook.
 * Part 2 (rth_*) : host scene layer -- C view of the C++ mirror of the reference's scene /
 *                  camera / config API (raytracing_amd/host/raytracing.hpp), which builds
 *                  the BVH on the host and flattens to rt_scene_desc.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ ABI handshake */
/* The number changes whenever a struct of this header changes its layout or a function its meaning.
 * A binding compares rt_abi_version() with the RT_ABI_VERSION it was built against before anything
 * else.  The three structs that have grown over time (rt_render_params, rt_stats, rt_accel_info)
 * also start with `struct_size`, which the caller sets to its own sizeof of the struct: fields are
 * only ever appended, so the library reads / writes the part both sides know and refuses what it
 * cannot interpret (struct_size 0, or parameters from a newer header than the library's). */
#define RT_ABI_VERSION 3u
uint32_t rt_abi_version(void);

/* ------------------------------------------------------------------ flat scene layout */

/* 32-bit reference = kind << 29 | index */
#define RT_REF_KIND_SHIFT 29u
#define RT_REF_INDEX_MASK 0x1FFFFFFFu
#define RT_KIND_NODE 0u   /* index into nodes[] */
#define RT_KIND_QUAD 1u   /* index into quads[] (parallelogram / triangle / disc) */
#define RT_KIND_SPHERE 2u /* index into spheres[] */
#define RT_KIND_MEDIUM 3u /* index into media[] */
#define RT_REF_NONE 0xFFFFFFFFu

/* BVH node, 64 B.  Interior: boxes of both children + child refs (any kind).
 * Leaf list (right == RT_NODE_LEAF): refs[left .. left+count) are scanned in order, which is
 * the order HittableVec::hit visits them (hittable.rs:252-264). */
#define RT_NODE_LEAF 0xFFFFFFFFu
typedef struct {
    float lmin[3];
    uint32_t left;
    float lmax[3];
    uint32_t right;
    float rmin[3];
    uint32_t count;
    float rmax[3];
    uint32_t pad;
} rt_node;

/* Quad-family primitive, 96 B (hittable.rs:281-677 with Translate/RotateY baked in).
 * avec = v x w and bvec = w x u so that alpha = (p-Q).avec, beta = (p-Q).bvec
 * (same value as w.((p-Q) x v) and w.(u x (p-Q)) at hittable.rs:357-358). */
#define RT_SHAPE_PARALLELOGRAM 0u
#define RT_SHAPE_TRIANGLE 1u
#define RT_SHAPE_DISC 2u
typedef struct {
    float q[3], d;      /* corner, plane constant D = normal . Q */
    float n[3], area;   /* unit normal, |u x v| */
    float avec[3];
    uint32_t shape;
    float bvec[3];
    uint32_t mat;
    float u[3];
    uint32_t prim_id;
    float v[3];
    uint32_t pad;
} rt_quad;

/* Sphere, 48 B (hittable.rs:117-209): centre(time) = c0 + time * cv.
 * u_shift: a sphere under RotateY keeps the texture longitude of its own frame
 * (Sphere::get_uv runs before the normal is rotated back, hittable.rs:194, 801-805), so
 * u = frac(u_world + u_shift) with u_shift = -angle / 2pi of the baked rotation. */
typedef struct {
    float c0[3], radius;
    float cv[3];
    uint32_t mat;
    uint32_t prim_id, moving;
    float u_shift;
    uint32_t pad;
} rt_sphere;

/* ConstantMedium (hittable.rs:820-891): boundary = refs[first .. first+count), a convex list */
typedef struct {
    uint32_t first, count;
    float neg_inv_density;
    uint32_t mat;
    uint32_t prim_id, pad0, pad1, pad2;
} rt_medium;

#define RT_MAT_LAMBERTIAN 0u
#define RT_MAT_LIGHT 1u
#define RT_MAT_METAL 2u
#define RT_MAT_DIELECTRIC 3u
#define RT_MAT_ISOTROPIC 4u
typedef struct {
    uint32_t type, tex; /* tex unused for metal / dielectric */
    float albedo[3];
    float param;        /* fuzz or refraction index */
    uint32_t needs_uv, pad;
} rt_material;

#define RT_TEX_SOLID 0u
#define RT_TEX_CHECKER 1u
#define RT_TEX_IMAGE 2u
#define RT_TEX_NOISE 3u
typedef struct {
    uint32_t type;
    float color[3];    /* solid */
    float scale;       /* checker: 1/scale as the reference multiplies (texture.rs:61); noise: scale */
    uint32_t a, b;     /* checker: even / odd texture ids; image: width / height */
    uint32_t offset;   /* image: byte offset into texels; noise: index of its table */
} rt_texture;

#define RT_LIGHTS_NONE 0u
#define RT_LIGHTS_SINGLE 1u /* `lights` is one primitive: no selection draw */
#define RT_LIGHTS_LIST 2u   /* `lights` is a list: uniform pick, averaged pdf */
typedef struct {
    uint32_t n_nodes, n_refs, n_quads, n_spheres, n_media, n_materials, n_textures, n_lights;
    uint32_t n_texel_bytes, n_noise; /* noise tables: n_noise * (256*4 floats + 3*256 ints) */
    uint32_t root;                   /* ref of the world */
    uint32_t lights_kind;
    uint32_t n_prims;                /* number of user-visible primitive ids */
    uint32_t max_stack;              /* traversal stack depth this scene needs */
    const rt_node *nodes;
    const uint32_t *refs;
    const rt_quad *quads;
    const rt_sphere *spheres;
    const rt_medium *media;
    const rt_material *materials;
    const rt_texture *textures;
    const uint32_t *light_refs;
    const uint8_t *texels;
    const float *noise_vec;   /* n_noise * 256 * 4 */
    const int32_t *noise_perm; /* n_noise * 3 * 256 */
} rt_scene_desc;

/* Outputs of Camera::build (camera.rs:280-366), computed on the host in f64. */
typedef struct {
    uint32_t image_width, image_height;
    uint32_t sqrt_spp;     /* floor(sqrt(samples_per_px)), camera.rs:341 */
    uint32_t max_depth;
    uint32_t background_sky; /* 1 = Background::Sky (camera.rs:473-482), 0 = Constant */
    uint32_t antialiasing;   /* RT_AA_SQUARE | RT_AA_DISC (AntialiasingType, camera.rs:204-211) */
    double background[3];
    double center[3], pixel_00[3], pxdelta_u[3], pxdelta_v[3];
    double defocus_angle;
    double defocus_disk_u[3], defocus_disk_v[3];
} rt_camera;

/* AntialiasingType (camera.rs:204-211).  The reference stores it and never reads it
 * (camera.rs:237): get_ray always samples the stratified square (camera.rs:409-414).  Here
 * Square is that; Disc does what the reference documents for it -- points from an r = 0.5 px
 * disc about the pixel centre, drawn by rejection like two_d.rs:183-190, not stratified. */
#define RT_AA_SQUARE 0u
#define RT_AA_DISC 1u

#define RT_MODE_REF_EXACT 0u /* estimator exactly as shipped, camera.rs:455-470 */
#define RT_MODE_BOOK3 1u     /* mixture estimator built from pdf.rs/onb.rs, specular bypass */

#define RT_FLAG_TIME_KERNELS 1u /* bracket kernels with HIP events (see rt_stats) */
#define RT_FLAG_WAVEFRONT 2u    /* ray-queue pipeline (scan/generate/extend/shade launches) instead of
                                 * the persistent path kernel; the same samples for the same seed (images equal
                                 * up to the order of the film's fp32 additions, as between any two renders) */
/* The persistent kernel has two variants.  Lane-bound: every lane carries one path from its camera
 * ray to its end.  Regrouping: every wave keeps a pool of paths in LDS and regroups them by the step
 * they wait for (closest hit, scatter at a diffuse / metal / glass vertex, end), so that each step runs
 * with most lanes active; behind a tree a lane keeps a traversal across turns and takes the next ray
 * from the pool when it is done.  The same samples either way (the film's fp32 sums are added in
 * whatever order the waves finish: two renders agree to the last bits, not bit for bit).  Default: regrouping on a flat world
 * (measured 40 % faster on the Book-3 Cornell box, 3 % on the main.rs scene) and behind trees of 8192
 * nodes or more (14 % faster on the 1 M-primitive scene, 2 % slower on the Book-1 scene's 232 nodes);
 * RT_FLAG_REGROUP asks for regrouping wherever it exists, RT_FLAG_NO_REGROUP for the lane-bound kernel.
 * Regrouping needs max_depth <= 255 and falls back otherwise. */
#define RT_FLAG_REGROUP 4u
#define RT_FLAG_NO_REGROUP 8u
/* The persistent kernels are compiled several times: for any scene, and without the code for what a
 * scene may lack (motion, media and textures behind a tree; in a flat world also every loop of the
 * scan but those over rectangle, upright-box and sphere records).  rt_render takes the variant that
 * fits the uploaded scene -- the same samples and the same arithmetic, fewer registers.
 * RT_FLAG_GENERAL_KERNELS takes the kernels compiled for any scene (tests compare the two). */
#define RT_FLAG_GENERAL_KERNELS 16u
/* Which compilation of the persistent kernel the last rt_render on this handle took (0 before any, and
 * after a wavefront render).  A diagnostic without a counterpart in the reference. */
#define RT_VARIANT_REGROUPED 1u /* paths regrouped per wave (rt_stats.regrouped) */
#define RT_VARIANT_TIMELESS 2u  /* flat world, nothing moves: pool slots without the ray's time */
#define RT_VARIANT_BOXED 4u     /* flat world of rectangle, upright-box and sphere records only */
#define RT_VARIANT_PLAIN 8u     /* behind a tree: no motion, no medium, no texture but colours */
#define RT_VARIANT_EXTRAS 16u   /* behind a tree: box leaves or primitives tested after the walk */
#define RT_VARIANT_SIMPLE 32u   /* lane-bound kernel with RT_VARIANT_EXTRAS: pinhole camera, square filter, one light */
typedef struct {
    uint32_t struct_size;                /* = sizeof(rt_render_params) of the caller's header */
    uint32_t mode;
    uint32_t flags;
    uint32_t reserved;                   /* 0 */
    uint64_t seed;
    uint32_t stratum_begin, stratum_end; /* stratum indices sj*sqrt_spp+si; end = 0 means sqrt_spp^2 */
    uint32_t pool_size;                  /* wavefront: rays in flight; 0 = default */
    uint32_t time_every;                 /* wavefront + RT_FLAG_TIME_KERNELS: time every n-th iteration (0 = all) */
    uint32_t refill_min;                 /* persistent: idle lanes of a wave before it takes new samples; 0 = default */
    uint32_t waves_per_simd;             /* persistent: waves per SIMD to launch; 0 = all that fit (4) */
} rt_render_params;

typedef struct {
    uint32_t struct_size;  /* in: sizeof(rt_stats) of the caller's header; out: bytes the library filled */
    uint32_t reserved;
    uint64_t segments;     /* closest-hit queries issued (one world.hit at camera.rs:444 each) */
    uint64_t camera_rays;
    uint64_t dropped;      /* samples skipped because their pdf weight was not finite/positive */
    uint64_t iterations;   /* wavefront iterations of the last rt_render */
    uint64_t launches[5];  /* generate, extend, shade, scan, paths (the persistent kernel) */
    uint64_t timed_launches[5]; /* launches bracketed by HIP events (RT_FLAG_TIME_KERNELS) */
    double kernel_ms[5];   /* summed HIP-event time of the timed launches, per kernel */
    double render_ms;      /* HIP-event time of the last rt_render, first launch to last */
    uint64_t regrouped;    /* 1 if the last rt_render ran the regrouping variant of the persistent kernel */
} rt_stats;

typedef struct rt_handle rt_handle;

/* ------------------------------------------------------------------ Part 1: device backend */
int rt_device_count(int *n);
int rt_create(int device_id, rt_handle **out);
void rt_destroy(rt_handle *h);
const char *rt_last_error(rt_handle *h); /* h may be NULL: error of the last failed rt_create */

/* Run all kernels of this handle on an existing hipStream_t (NULL = the handle's own). */
int rt_set_stream(rt_handle *h, void *hip_stream);
int rt_synchronize(rt_handle *h);

/* Copies the flat scene to HBM.  Replaces the Rc<dyn Hittable> graph handed to
 * Camera::render (camera.rs:368). */
int rt_upload_scene(rt_handle *h, const rt_scene_desc *scene);

/* Acceleration structure built by the next rt_upload_scene.  The descriptor always carries the
 * reference's own grouping (BVHNode::new topology, boundingbox.rs:179-206, and HittableVec
 * lists); the closest hit does not depend on the grouping, so the device may regroup:
 *   width 4 (default, with RT_BUILD_AUTO): binary nodes and lists collapsed into 4-wide nodes with boxes recomputed
 *           from the fp32 primitives; builder RT_BUILD_REFERENCE keeps the descriptor's
 *           grouping, RT_BUILD_SAH regroups all primitives by a binned surface-area heuristic,
 *           RT_BUILD_AUTO builds both and keeps the one that is cheaper to walk (rt_accel_info.walk_cost);
 *   width 2: the descriptor's nodes traversed as they are (near child first).
 * A world that is one short list of primitives is scanned without any tree either way; boxes
 * at the end of its quads (main.rs:52-69) are recognised and tested as boxes, with the hit
 * reported on the face's own quad. */
#define RT_BUILD_REFERENCE 0u
#define RT_BUILD_SAH 1u
#define RT_BUILD_AUTO 2u /* both, keep the one that is cheaper to walk */
/* Or-ed into the builder: the size of a 4-wide node.  128 bytes with fp32 child boxes, or 64 bytes
 * whose child boxes are bytes on a grid of the node's own, rounded outward -- the same tree and the
 * same hits, half the bytes and four loads instead of seven per visit, a little more arithmetic.
 * With neither flag the upload picks by the size of the tree (byte boxes from 8192 nodes). */
#define RT_BUILD_FP32_BOXES 0x100u
#define RT_BUILD_BYTE_BOXES 0x200u
/* Or-ed into the builder: six parallelograms of one list that are the faces of a box along the
 * world's axes (box3, hittable.rs:398-455) normally become ONE leaf of the 4-wide tree, searched
 * with one slab test and reported on the face's own quad -- same hits, same primitive ids, a sixth
 * of the leaves.  This flag keeps them six quads (for comparisons). */
#define RT_BUILD_NO_BOX_LEAVES 0x400u
/* Or-ed into the builder: a sphere or medium that encloses the rest of the scene (the fog of radius
 * 5000 around "The Next Week"'s final scene) is normally kept out of the 4-wide tree and tested once
 * per ray when the ray's walk through the tree has ended -- no box can cull it, and the closest hit
 * does not depend on the order of the tests.  This flag leaves such primitives in the tree. */
#define RT_BUILD_NO_POST_TESTS 0x800u
int rt_set_accel(rt_handle *h, uint32_t builder, uint32_t width);
typedef struct {
    uint32_t struct_size;   /* in: sizeof(rt_accel_info) of the caller's header; out: bytes the library filled */
    uint32_t kind;          /* 0 = flat list scan, 1 = binary nodes, 2 = 4-wide nodes */
    uint32_t nodes, depth;  /* of the structure searched (depth: wide only) */
    uint32_t stack_entries; /* per-lane traversal stack held in LDS */
    double cost;            /* wide only: expected box tests of a random ray through the root box */
    uint32_t flat_quads;    /* flat only: quads tested one by one */
    uint32_t flat_boxes;    /* flat only: runs of six quads at the end of the list that are the faces of one
                             * rectangular box (box3, hittable.rs:398-455), and the five or six walls of a
                             * room the list starts with; they take one slab test each */
    uint32_t flat_alone_spheres; /* flat only: spheres that no other primitive of the list reaches into; the
                             * regrouping kernel knows where a ray going into one of them ends */
    uint32_t node_bytes;    /* size of a node of the structure searched (wide: 64 quantised, 128 fp32 boxes) */
    uint32_t flat_boxes_aligned; /* of flat_boxes: along the three world axes (the walls of a room) */
    uint32_t flat_boxes_upright; /* of flat_boxes: one axis the world's y, as under RotateY and Translate
                             * (hittable.rs:760-1001); the slab test skips the dot products such axes
                             * do not need.  Both are 0 when the list orders its boxes otherwise. */
    uint32_t flat_rects;    /* flat only: rectangles along the world's axes (a lamp under a ceiling) that left
                             * flat_quads for a test without dot products; same hits, same tie order */
    uint32_t box_leaves;    /* wide only: leaves that are a whole axis-aligned box (six quads each) */
    uint32_t leaves;        /* wide only: leaves of the tree (a box leaf counts once) */
    uint32_t post_tests;    /* wide only: primitives enclosing the scene, tested after the walk instead of in the tree */
    uint32_t reserved;
    double walk_cost;       /* wide only: measured work of walking a surface-to-surface segment through the tree, in
                             * node visits; what RT_BUILD_AUTO compares */
} rt_accel_info;
int rt_get_accel_info(rt_handle *h, rt_accel_info *out);
/* Host only (no device needed): builds the 4-wide structure for a descriptor and checks it
 * against the descriptor -- same leaves, boxes that contain their content, forward indices.
 * stack_entries reports the exact stack need; the flat_* fields say what the upload would make
 * of the scene as a flat world (all 0 unless the world is one list of primitives).  Returns 0,
 * or < 0 with the reason in err. */
int rt_accel_selftest(const rt_scene_desc *scene, uint32_t builder, rt_accel_info *out, char *err, uint32_t err_len);

/* The pixel x stratum loop of camera.rs:386-401 for strata [begin,end): every sample's
 * radiance is ADDED into the film (planar fp32 sums in HBM).  Asynchronous on the stream
 * except for the completion polling it does itself; returns when all paths have finished. */
int rt_render(rt_handle *h, const rt_camera *cam, const rt_render_params *params);
/* RT_VARIANT_* bits of the persistent kernel the last rt_render on this handle took (a diagnostic, no
 * counterpart in the reference; h may be NULL: 0). */
uint32_t rt_last_kernel_variant(const rt_handle *h);

/* Film = 3 planes of width*height fp32 SUMS (R plane, G plane, B plane). */
int rt_film_clear(rt_handle *h);
int rt_film_resize(rt_handle *h, uint32_t width, uint32_t height); /* allocate + clear */
int rt_film_bind(rt_handle *h, void *device_ptr, uint32_t width, uint32_t height); /* external */
int rt_film_device_ptr(rt_handle *h, void **device_ptr, uint64_t *n_floats);
/* rgb: width*height*3 interleaved, each = sum * scale (scale = 1/spp gives camera.rs:397) */
int rt_film_download(rt_handle *h, float *rgb, float scale);
/* Restores film sums saved by rt_film_download(h, rgb, 1.0f) (interleaved, width*height*3):
 * with the stratum range of rt_render_params this resumes an interrupted render. */
int rt_film_upload(rt_handle *h, const float *rgb, uint32_t width, uint32_t height);
/* Output stage on the device: rgb8 (width*height*3 bytes, host) = the reference's 8-bit rule on
 * sum * scale -- `pixel_color * px_sample_scale` (camera.rs:397), then as_gamma_corrected
 * (gamma != 0; PpmWriter, export.rs:35-36, color.rs:131-137, 209-215) or linear (PngWriter,
 * export.rs:80), then as_rgb_ints: clamp to [0,1], * 255, truncate (color.rs:148-155).
 * Evaluated in f64 on the fp32 sums. */
int rt_film_quantize(rt_handle *h, uint8_t *rgb8, double scale, uint32_t gamma);
/* rt_film_quantize for callers that cannot allocate.  No counterpart in the reference: it is for
 * tight serving / preview loops and harnesses that forbid hipMalloc once they run.  Same
 * arguments, same bytes.  The packed image is staged in a fixed device array that every GPU
 * holds from library load to process exit (24.9 MB), so the film may have at most
 * RT_NOALLOC_MAX_PIXELS pixels (3840 x 2160); a larger film is refused with an error.  The
 * library itself makes no hipMalloc / hipFree or other allocating call here; the copy to `rgb8`
 * is a device-to-host hipMemcpyAsync, so pass pinned memory (hipHostMalloc) if the HIP runtime
 * must not stage or pin anything on its own either.  Like rt_film_quantize it blocks until the
 * bytes have arrived (a stream synchronise): it cannot be called while the handle's stream is
 * being captured into a HIP graph.  All handles of one GPU share the array; calls on one GPU are
 * serialised by a per-GPU lock inside the library, and the array keeps the last call's bytes
 * (every word the copy reads is overwritten first, so nothing needs clearing). */
#define RT_NOALLOC_MAX_PIXELS 8294400u
int rt_film_quantize_noalloc(rt_handle *h, uint8_t *rgb8, double scale, uint32_t gamma);
int rt_get_stats(rt_handle *h, rt_stats *out);

/* ------------------------------------------------------------------ several GPUs: the film reduce
 * Camera::render (camera.rs:368) sums `sqrt_spp^2` strata per pixel (camera.rs:390-395); the strata
 * are independent, so each GPU renders the full frame for its own stratum range (rt_render_params)
 * and the films -- fp32 SUMS -- are added once: an RCCL all-reduce (sum, fp32, width*height*3 values)
 * enqueued on each handle's stream, behind that handle's render.  The reference has no counterpart
 * (it is one thread).  librccl.so.1 is loaded when the first of these calls is made, not before.
 *
 * Two ways to form the group:
 *   one process, one handle per GPU:   rt_comm_init_all(handles, n)      (ncclCommInitAll)
 *   one process per GPU (n ranks):     rank 0 calls rt_comm_unique_id(id) and hands the
 *       RT_COMM_ID_BYTES bytes to its peers by any means; every rank then calls
 *       rt_comm_init_rank(h, id, rank, n)                                 (ncclCommInitRank)
 * A handle belongs to at most one group; rt_comm_destroy (or rt_destroy) leaves it.
 *
 * rt_film_allreduce(handles, n): every film becomes the sum of all films of the group.  `handles`
 * are this process's members of the group -- all n of them after rt_comm_init_all, the one local
 * handle (n = 1) after rt_comm_init_rank; the films must have equal dimensions.  Asynchronous:
 * the sums are complete once each handle's stream has passed the call (rt_synchronize, or any
 * later call on the handle that reads the film).  A group of one rank returns the film unchanged. */
#define RT_COMM_ID_BYTES 128u
int rt_comm_init_all(rt_handle *const *handles, int n);
int rt_comm_unique_id(void *id /* RT_COMM_ID_BYTES */, char *err, uint32_t err_len);
int rt_comm_init_rank(rt_handle *h, const void *id /* RT_COMM_ID_BYTES */, int rank, int n_ranks);
int rt_comm_destroy(rt_handle *h);
/* rank of the handle in its group and the group's size; both 0 when it is in none */
int rt_comm_info(rt_handle *h, int *rank, int *n_ranks);
int rt_film_allreduce(rt_handle *const *handles, int n);

/* Parity hook: closest hit of the uploaded world for n host rays (o,d: n*3; time: n or NULL).
 * origin_prim (n or NULL): id of the primitive each ray starts on, -1 for none -- what the
 * shade pass records for every bounce ray, so that the surface a ray leaves is not met again
 * at t ~ 0 (the reference relies on its 0.001 lower bound in f64 for that).
 * prim = user-visible primitive id or -1.  ConstantMedium draws are keyed by ray index. */
int rt_intersect_batch(rt_handle *h, uint32_t n, const float *o, const float *d, const float *time,
                       const int32_t *origin_prim, float tmin, float tmax, int32_t *hit, int32_t *prim,
                       float *t, float *u, float *v);
/* Camera rays exactly as the generate kernel makes them (camera.rs:409-430). */
int rt_camera_rays(rt_handle *h, const rt_camera *cam, const rt_render_params *params, uint32_t n,
                   const uint32_t *pixel, const uint32_t *stratum, float *o, float *d, float *time);
/* Device-side unit hooks for the per-function parity tests. */
int rt_eval_texture(rt_handle *h, uint32_t tex, uint32_t n, const float *uvp /* n*5: u v px py pz */,
                    float *rgb);
int rt_philox(rt_handle *h, uint32_t n, const uint32_t *ctr /* n*4 */, const uint32_t key[2],
              uint32_t *out /* n*4 */);

/* ------------------------------------------------------------------ Part 2: host scene layer */
typedef struct rth_scene rth_scene;

typedef struct { /* CameraBuilder (camera.rs:14-181) */
    uint32_t image_width, image_height;
    double vfov_deg;
    uint32_t samples_per_px;
    uint32_t max_depth;
    int32_t background_sky;
    int32_t antialiasing; /* RT_AA_SQUARE (default) | RT_AA_DISC, CameraBuilder::antialias (camera.rs:114-123) */
    double background[3];
    double center[3], target[3], vup[3];
    double defocus_angle_deg;
    double focal_length;
} rth_camera_builder;

void rth_camera_defaults(rth_camera_builder *b); /* camera.rs:183-202 */
/* with_aspect_ratio (camera.rs:78-92): sets width and round(width / aspect) */
int rth_camera_with_aspect_ratio(rth_camera_builder *b, uint32_t width, double aspect_ratio);
/* CameraBuilder::build + Camera::build.  On validation failure returns <0 and writes the
 * newline-joined "CameraBuilder::..." messages (camera.rs:53-55) into err. */
int rth_camera_build(const rth_camera_builder *b, rt_camera *out, char *err, uint32_t err_len);

rth_scene *rth_scene_new(void);
void rth_scene_free(rth_scene *);
const char *rth_last_error(rth_scene *);

int rth_tex_solid(rth_scene *, double r, double g, double b);
int rth_tex_checker(rth_scene *, double scale, int even, int odd);
int rth_tex_image(rth_scene *, uint32_t w, uint32_t h, const uint8_t *rgb);
int rth_tex_image_png(rth_scene *, const char *path); /* 8-bit RGB PNG only (texture.rs:95-122) */
int rth_tex_noise(rth_scene *, double scale, const double *ranvec, const int32_t *perm);

int rth_mat_lambertian(rth_scene *, int tex);
int rth_mat_light(rth_scene *, int tex);
int rth_mat_metal(rth_scene *, double r, double g, double b, double fuzz);
int rth_mat_dielectric(rth_scene *, double ior);
int rth_mat_isotropic(rth_scene *, int tex);

int rth_sphere(rth_scene *, const double c0[3], const double cv[3], double radius, int mat);
int rth_quad(rth_scene *, int shape, const double q[3], const double u[3], const double v[3], int mat);
int rth_box3(rth_scene *, const double a[3], const double b[3], int mat);
int rth_list(rth_scene *, const int *objs, int n);
int rth_bvh(rth_scene *, const int *objs, int n); /* BVHNode::new, boundingbox.rs:179-206 */
int rth_translate(rth_scene *, int obj, const double off[3]);
int rth_rotate_y(rth_scene *, int obj, double angle_rad);
int rth_medium(rth_scene *, int boundary, double density, int tex);
int rth_set_world(rth_scene *, int obj);
int rth_set_lights(rth_scene *, int obj);
int rth_prim_count(rth_scene *);

/* ConfigModel::from_str + as_world (config.rs:628-706): parses the TOML text, adds its
 * textures / materials / objects to the scene and returns the object id of the world
 * HittableVec (>= 0), or < 0 with the message in rth_last_error. */
int rth_load_toml(rth_scene *, const char *toml_text);
/* A scene file: the same format extended to a whole render -- a [camera] table with the
 * CameraBuilder fields (camera.rs:134-171), [r, g, b] colours, Box / ConstantMedium / Group
 * objects, center_vec, rotate_y / translate, `light = true` (see raytracing_amd/host/config.cpp).
 * Sets the scene's world and lights (the two arguments of Camera::render, camera.rs:368); image
 * paths are relative to base_dir.  camera receives the builder settings (defaults where the file
 * is silent), has_camera whether the file has a [camera] table.  Returns the world's object id. */
int rth_load_scene_toml(rth_scene *, const char *toml_text, const char *base_dir, rth_camera_builder *camera,
                        int *has_camera);
int rth_list_add(rth_scene *, int list, int obj); /* HittableVec::add, hittable.rs:238-244 */
int rth_material_by_name(rth_scene *, const char *name); /* materials loaded from TOML */

/* Flatten world + lights into the arrays of rt_scene_desc (owned by the scene; valid until
 * the next flatten or rth_scene_free). */
int rth_flatten(rth_scene *, const rt_scene_desc **out);
int rth_bvh_stats(rth_scene *, int obj, int *nodes, int *depth);

/* Host-side pure functions kept for the reference's known-answer tests. */
void rth_color_hex(uint32_t hex, double rgb[3]);                         /* color.rs:46-64 */
void rth_color_quantize(const double rgb[3], int gamma, uint8_t out[3]); /* color.rs:130-155 */
void rth_sphere_uv(const double p[3], double uv[2]);                     /* hittable.rs:150-162 */
int rth_interval_contains(double a, double b, double x);
int rth_interval_surrounds(double a, double b, double x);
int rth_interval_overlap(double a0, double a1, double b0, double b1, double out[2]);
void rth_interval_expand(double a, double b, double delta, double out[2]);
double rth_interval_clamp(double a, double b, double x);
/* ImageWriter (export.rs): 8-bit files from width*height*3 interleaved linear floats.
 * PNG writes linear values, PPM applies gamma 2, both truncate (color.rs:148-155). */
int rth_write_png(const char *path, uint32_t w, uint32_t h, const float *rgb);
int rth_write_ppm(const char *path, uint32_t w, uint32_t h, const float *rgb);
/* The same files from bytes already quantised (rt_film_quantize with gamma 0 / 1). */
int rth_write_png_rgb8(const char *path, uint32_t w, uint32_t h, const uint8_t *rgb8);
int rth_write_ppm_rgb8(const char *path, uint32_t w, uint32_t h, const uint8_t *rgb8);
/* 8-bit RGB PNG -> bytes (the loader behind ImageTexture, texture.rs:95-122); rgb8 may be NULL
 * to query the size. */
int rth_read_png_rgb8(const char *path, uint32_t *w, uint32_t *h, uint8_t *rgb8, uint64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
