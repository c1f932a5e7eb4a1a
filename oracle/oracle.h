/*
 * oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a double-precision, single-threaded,
 * recursive restatement of the reference renderer (piemot/raytracing, Rust) and exists so
 * that tests, the smoke check and the benchmark's cpu_baseline leg have something to compare
 * the HIP path against.  Nothing in the product (raytracing_amd/, include/) links, loads or
 * calls it.
 *
 * Parity pin: the reference cannot be compiled in this environment (no Rust toolchain), so
 * the restatement is pinned only by the reference's own known-answer tests for this path
 * (hittable.rs:896-909 sphere uv, interval.rs doc-tests) and by hand-derived closed forms.
 * Everything the reference leaves as todo!()/unimplemented!() (specular materials under the
 * PDF estimator, sphere/list lights, the mixture estimator wiring, Perlin noise) follows
 * "Ray Tracing: The Rest of Your Life" and is PARITY UNPINNED.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* estimator modes */
#define ORC_MODE_REF_EXACT 0 /* camera.rs:432-471 as shipped: light-only sampling */
#define ORC_MODE_BOOK3 1     /* mixture estimator, specular bypass */

/* random streams */
#define ORC_RNG_PHILOX24 0 /* Philox4x32-10, 24-bit uniforms: the stream the device uses */
#define ORC_RNG_PHILOX53 1 /* same counters, 53-bit uniforms */
#define ORC_RNG_MT64 2     /* mt19937_64, sequential: an independent stream */

/* quad-family shapes (hittable.rs:327-337, 509-517, 627-638) */
#define ORC_SHAPE_PARALLELOGRAM 0
#define ORC_SHAPE_TRIANGLE 1
#define ORC_SHAPE_DISC 2

typedef struct {
    uint32_t image_width, image_height;
    double vfov_deg;
    uint32_t samples_per_px; /* requested; floor(sqrt)^2 is rendered (camera.rs:341-342) */
    uint32_t max_depth;
    int32_t background_sky; /* 1 = Background::Sky, 0 = Constant(background) */
    int32_t antialiasing;   /* 0 = Square: what get_ray does (camera.rs:409-414); 1 = Disc: r = 0.5 px disc
                             * about the pixel centre as AntialiasingType documents it (camera.rs:204-211),
                             * which the reference never implemented -- parity unpinned */
    double background[3];
    double center[3], target[3], vup[3];
    double defocus_angle_deg;
    double focal_length;
} orc_camera;

typedef struct {
    int32_t mode;
    int32_t rng_kind;
    uint64_t seed;
    uint32_t stratum_begin, stratum_end; /* range of stratum indices sj*sqrt_spp+si; end=0 => all */
    int32_t sanitize;                    /* 1: drop samples whose pdf weight is not finite/positive */
    int32_t threads;                     /* <=1: single thread (like the reference); >1 OpenMP rows */
    uint32_t x0, y0, x1, y1;             /* pixel window [x0,x1) x [y0,y1); x1=0 => full frame */
} orc_render_params;

/* counters: [0]=segments [1]=box tests [2]=sphere tests [3]=sphere accepts [4]=quad plane tests
 * [5]=quad interior tests [6]=dropped samples [7]=camera rays
 * [8]=segments traced below a vertex whose weight was exactly 0 (they cannot change the image) */
#define ORC_NCOUNTERS 10

orc_scene *orc_scene_new(void);
void orc_scene_free(orc_scene *);
const char *orc_last_error(orc_scene *);

/* textures -> texture id (>=0) or -1 */
int orc_tex_solid(orc_scene *, double r, double g, double b);
int orc_tex_checker(orc_scene *, double scale, int even, int odd);
int orc_tex_image(orc_scene *, uint32_t w, uint32_t h, const uint8_t *rgb);
int orc_tex_noise(orc_scene *, double scale, const double *ranvec /*256*3*/, const int32_t *perm /*3*256*/);

/* materials -> material id */
int orc_mat_lambertian(orc_scene *, int tex);
int orc_mat_light(orc_scene *, int tex);
int orc_mat_metal(orc_scene *, double r, double g, double b, double fuzz);
int orc_mat_dielectric(orc_scene *, double ior);
int orc_mat_isotropic(orc_scene *, int tex);

/* objects -> object id.  Leaf primitives get consecutive primitive ids in creation order. */
int orc_sphere(orc_scene *, const double c0[3], const double cv[3], double radius, int mat);
int orc_quad(orc_scene *, int shape, const double q[3], const double u[3], const double v[3], int mat);
int orc_box3(orc_scene *, const double a[3], const double b[3], int mat);
int orc_list(orc_scene *, const int *objs, int n);
int orc_bvh(orc_scene *, const int *objs, int n);
int orc_translate(orc_scene *, int obj, const double off[3]);
int orc_rotate_y(orc_scene *, int obj, double angle_rad);
int orc_medium(orc_scene *, int boundary, double density, int tex);
int orc_set_world(orc_scene *, int obj);
int orc_set_lights(orc_scene *, int obj); /* -1: no lights */
int orc_prim_count(orc_scene *);

/* film_sum: W*H*3 doubles (sum over strata, not divided); film_sumsq may be NULL. */
int orc_render(orc_scene *, const orc_camera *, const orc_render_params *, double *film_sum,
               double *film_sumsq, uint64_t *counters);

/* The same render, and per pixel how many of its samples' paths scattered off a sphere of radius <
 * radius_below on their way (any material: the outgoing direction is built on the normal (p - c) / r,
 * which carries a difference in the last bit of the hit point p multiplied by 1 / r), so these are the
 * paths on which an fp32 render and this f64 one part ways.  flagged: W*H counts. */
int orc_render_flagged(orc_scene *, const orc_camera *, const orc_render_params *, double *film_sum, double radius_below,
                       uint32_t *flagged);

/* camera rays as the device generates them (for fixed ray batches) */
int orc_camera_rays(orc_scene *, const orc_camera *, const orc_render_params *, uint32_t n,
                    const uint32_t *pixel, const uint32_t *stratum, double *o, double *d, double *time);

/* closest hit of world for a batch. second_t: t of the runner-up primitive (inf if none). */
int orc_intersect_batch(orc_scene *, uint32_t n, const double *o, const double *d, const double *time,
                        double tmin, double tmax, int32_t *hit, int32_t *prim, double *t, double *u,
                        double *v, double *second_t, double *normal /* n*3 or NULL */);

/* per-function hooks */
void orc_sphere_uv(const double p[3], double uv[2]);            /* hittable.rs:150-162 */
void orc_onb(const double n[3], double uvw[9]);                 /* onb.rs:13-28 */
void orc_cosine_sample(double r1, double r2, double out[3]);    /* vec.rs:246-256 */
void orc_reflect(const double v[3], const double n[3], double out[3]);
void orc_refract(const double uv[3], const double n[3], double ratio, double out[3]); /* vec.rs:289-295 */
double orc_schlick(double cosine, double ri);                   /* material.rs:146-151 */
void orc_color_hex(uint32_t hex, double rgb[3]);                /* color.rs:46-64 */
void orc_color_quantize(const double rgb[3], int gamma, uint8_t out[3]); /* color.rs:130-155 */
int orc_interval_contains(double a, double b, double x);        /* interval.rs:86-88 */
int orc_interval_surrounds(double a, double b, double x);       /* interval.rs:114-116 */
int orc_interval_overlap(double a0, double a1, double b0, double b1, double out[2]); /* :44-54 */
void orc_interval_expand(double a, double b, double delta, double out[2]);          /* :151-153 */
double orc_interval_clamp(double a, double b, double x);        /* :130-138 */
void orc_tex_value(orc_scene *, int tex, double u, double v, const double p[3], double rgb[3]);
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_philox2x32(const uint32_t ctr[2], uint32_t key, uint32_t out[2]); /* the media's generator */
int orc_aabb_hit(const double bmin[3], const double bmax[3], const double o[3], const double d[3],
                 double tmin, double tmax);                     /* boundingbox.rs:96-113 */
int orc_bvh_stats(orc_scene *, int obj, int *nodes, int *depth);

#ifdef __cplusplus
}
#endif
#endif
