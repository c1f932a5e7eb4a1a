// rt_kernels.h -- path-tracing kernels for gfx950 (fp32).
//
// Replaces the body of Camera::render (camera.rs:386-401) and everything below it.  Two
// pipelines share the intersection, camera, light and material code in this file:
//   * k_paths (default): one persistent launch per frame; a lane carries one path in registers
//     from its camera ray to its end and then takes the next (pixel, stratum) sample.  No ray
//     state goes through memory; see the comment above k_paths.
//   * the ray-queue pipeline (RT_FLAG_WAVEFRONT), four kernels per iteration, described next.
//
// Ray-queue work layout: the ray pool is split into CHUNKS, one per wavefront.  A chunk holds up to
// `cap` rays, dense at its front; its wave walks it in tiles of 64.  Because a wave owns its
// chunk, stream compaction after shading is a ballot + mbcnt prefix inside the wave with no
// atomics and no barriers, and survivors are written in place (writes never pass the tile
// being read).  One iteration = scan -> generate -> extend -> shade:
//   scan     : one block; prefix-sums the free slots of all chunks and hands each chunk a
//              contiguous range of new (pixel, stratum) sample ids; bookkeeping counters.
//   generate : camera rays (camera.rs:409-430) appended to each chunk.
//   extend   : closest hit of the world (camera.rs:444).
//   shade    : emission, material scatter, light sampling, PDF weights, throughput update,
//              film accumulation (camera.rs:451-470), compaction of surviving rays.
#pragma once
#include "rt_math.h"
#include "rt_leafword.h"
#include <type_traits>
#include "../../include/rt_amd.h"

namespace rt {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;          // 4 waves = 4 chunks per block
constexpr int STACK_DEPTH = 32;     // per-lane traversal stack entries (LDS)
constexpr int LDS_NODES = 128;      // BVH nodes staged per block (8 KiB)
#ifndef RT_NODE_SHARE
#define RT_NODE_SHARE 2
#endif
constexpr uint32_t NODE_SHARE = RT_NODE_SHARE; // see trav4_round
#ifndef RT_SHADE_QUORUM
#define RT_SHADE_QUORUM 48
#endif
constexpr uint32_t SHADE_QUORUM = RT_SHADE_QUORUM; // see k_paths

// Read-only scene records through the constant address space: with a wave-uniform address the
// backend selects scalar loads (s_load_dwordx4..x16), the record lands in SGPRs and feeds the
// VALU directly -- no vector-memory instruction and no VGPRs per primitive.  Only for arrays
// nothing writes while a kernel runs.
#define RT_CONST_AS __attribute__((address_space(4)))
template <typename T> __device__ __forceinline__ const RT_CONST_AS T *as_uniform(const T *p) {
    return reinterpret_cast<const RT_CONST_AS T *>(reinterpret_cast<uintptr_t>(p));
}

// instrumented builds (make stats-lib): turns, lanes and clock ticks per step of the persistent kernels
constexpr uint32_t RG_STATS = 8;
#ifdef RT_REGROUP_STATS
__device__ unsigned long long g_rg_stats[3 * RG_STATS];
struct TravStats { // of the wide traversal, per wave
    unsigned long long node_steps = 0, node_lanes = 0, prim_steps = 0, prim_lanes = 0;
};
#else
struct TravStats {};
#endif

// LIGHTS template arguments: what the caller knows about DScene::lights_kind
constexpr uint32_t LIGHTS_ANY = 0xFFu;
#define RT_LKIND(sc) (LIGHTS == LIGHTS_ANY ? (sc).lights_kind : LIGHTS)
struct DScene {
    const rt_node *nodes;
    const uint32_t *refs;
    const float4 *quads;   // 6 float4 per quad
    const float4 *spheres; // 3 float4 per sphere
    const rt_medium *media;
    const rt_material *mats;
    const rt_texture *texs;
    const uint32_t *light_refs;
    const float4 *light_recs; // the light list's primitives, 8 float4 per light (see light_random)
    const uint8_t *texels;
    const float4 *noise_vec;
    const int *noise_perm;
    // a world that is one list of primitives: its quads packed in list order, 16 floats each
    // (q.xyz ref | n.xyz mat | avec.xyz shape | bvec.xyz -), and the refs of everything else
    const float *flat_quads;
    const float *flat_pairs; // the same quads two per record when all are parallelograms (see scan_flat)
    const uint32_t *flat_other;
    const float *flat_spheres; // when all of flat_other would be spheres: their records, 16 floats each (see scan_flat)
    // runs of six quads at the end of the list's quads that are the faces of one rectangular box
    // (box3, hittable.rs:398-455, under any rigid map): 32 floats each, see scan_flat
    const float *flat_boxes;
    // 4-wide BVH derived at upload (rt_accel.h), root = node 0: 8 float4 per node (fp32 boxes) or 4
    // (quantised boxes)
    const float4 *nodes4;
    const float4 *wide_boxes; // leaves of kind RT_KIND_BOX: 3 float4 per box (WideBox, rt_accel.h)
    uint32_t root, n_lights, lights_kind, n_nodes, n_flat_quads, n_flat_pairs, n_flat_other, n_nodes4, n_flat_boxes;
    uint32_t n_flat_boxes_aligned, n_flat_boxes_upright; // the box records' first two classes (scan_flat)
    // rectangles along the world's axes: 8 floats each at the head of flat_boxes, padded to 32; sorted
    // by normal axis: the first _x have it along x, the next _y along y, the rest along z
    uint32_t n_flat_rects, n_flat_rects_x, n_flat_rects_y;
    uint32_t n_flat_spheres;
    // dynamic LDS of the wide traversal: [stack4][BLOCK] stack entries, then the first lds_nodes4 nodes
    uint32_t stack4, lds_nodes4;
    // wide tree: primitives that enclose the rest of the scene (a fog around the world) are not in the
    // tree; every ray tests them when its walk has ended (test_post).  Leaf words, class bits included.
    uint32_t n_post, post_ref[2];
};

// which structure a kernel instance searches
// ACCEL_BVH4Q: the 4-wide tree with 64-byte nodes whose child boxes are 8-bit offsets on a grid of
// the node's own (see trav4_round); ACCEL_BVH4: the same tree with 128-byte fp32 nodes.
enum : int { ACCEL_FLAT = 0, ACCEL_BVH2 = 1, ACCEL_BVH4 = 2, ACCEL_BVH4Q = 3 };
constexpr bool accel_is_wide(int accel) { return accel == ACCEL_BVH4 || accel == ACCEL_BVH4Q; }

struct DCamera {
    V3 p00_rel, du, dv, center, disk_u, disk_v, bg;
    uint32_t W, H, npix, sqrt_spp, stratum_begin, max_depth, defocus, sky, aa_disc, timeless;
    float inv_sqrt, inv_w;
};

// Scene and camera live in device memory and are read through the constant address space:
// fields come in by scalar loads where they are used instead of occupying ~70 SGPRs for the
// whole kernel (as by-value kernel arguments they did, and spilled).
using SceneRef = const RT_CONST_AS DScene &;
using CamRef = const RT_CONST_AS DCamera &;
__device__ __forceinline__ V3 ldv(const RT_CONST_AS V3 &v) { return V3{v.x, v.y, v.z}; }

struct Ctrl {
    unsigned long long next_sample, total_samples, segments, camera_rays, dropped;
    uint32_t active, iterations;
};

// Ray queue, SoA of 16-byte records so every access is one dwordx4 per lane.
struct Queue {
    float4 *ra;  // origin.xyz, time
    float4 *rb;  // direction.xyz, ref of the primitive the ray leaves (bits)
    float4 *sa;  // throughput.rgb, pixel (bits)
    uint2 *sb;   // stratum, bounce
    float4 *hit; // t, ref (bits), u, v
    uint32_t *count;             // rays per chunk
    unsigned long long *gen_base; // first new sample id per chunk
    uint32_t *gen_n;             // new rays per chunk
    uint32_t cap, n_chunks;
};

__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __uint_as_float(u); }

// ------------------------------------------------------------------ primitives
struct Best {
    float t;
    uint32_t ref;
    float u, v; // surface coordinates; the flat scan leaves a quad's for quad_uv (below)
    uint32_t cls; // wide traversal only: RT4_CLASS_* of the hit's material (rt_accel.h)
};

// A 16-byte record element either per lane (vector load) or wave-uniform (scalar load).
template <bool UNIFORM> __device__ __forceinline__ float4 load4(const float4 *base, size_t i) {
    if (UNIFORM) {
        const RT_CONST_AS float *p = as_uniform(reinterpret_cast<const float *>(base)) + 4 * i;
        return make_float4(p[0], p[1], p[2], p[3]);
    }
    return base[i];
}

// Parallelogram / Triangle / Disc (hittable.rs:341-371, 520-551, 641-672).
// t in [tlo, thi] closed (Interval::contains).
template <bool UNIFORM = false>
__device__ __forceinline__ bool hit_quad(const float4 *__restrict__ Q, V3 o, V3 d, float tlo, float thi,
                                         float &t, float &a, float &b) {
    float4 q0 = load4<UNIFORM>(Q, 0), q1 = load4<UNIFORM>(Q, 1);
    V3 n = mk(q1);
    float denom = dot(n, d);
    if (fabsf(denom) < 1e-8f) return false;
    V3 qo = mk(q0) - o;
    // (D - n.o)/denom with the subtraction done first; v_rcp_f32 and a product (1.5 ulp) -- the same
    // two roundings hit_wide_box makes for a face along a world axis, so equal planes give equal t
    float tt = dot(n, qo) * rcp(denom);
    if (!(tt >= tlo && tt <= thi)) return false;
    float4 q2 = load4<UNIFORM>(Q, 2), q3 = load4<UNIFORM>(Q, 3);
    V3 ph = fma3(d, tt, -qo);
    float alpha = dot(ph, mk(q2));
    float beta = dot(ph, mk(q3));
    uint32_t shape = fbits(q2.w);
    bool inside;
    if (shape == RT_SHAPE_PARALLELOGRAM) {
        inside = (0.0f <= alpha && alpha <= 1.0f) && (0.0f <= beta && beta <= 1.0f);
    } else if (shape == RT_SHAPE_TRIANGLE) {
        inside = !(alpha < 0.0f || beta < 0.0f || (alpha + beta) > 1.0f);
    } else {
        float ac = fabsf(0.5f - alpha), bc = fabsf(0.5f - beta);
        inside = !(alpha < 0.0f || beta < 0.0f || sqrtf(ac * ac + bc * bc) > 0.5f);
    }
    if (!inside) return false;
    t = tt;
    a = alpha;
    b = beta;
    return true;
}

// (alpha, beta) of a point of the quad's plane (hittable.rs:355-358): what hit_quad computes on
// its way, for callers that have only the hit point.
__device__ __forceinline__ void quad_uv(const float4 *__restrict__ Q, V3 p, float &a, float &b) {
    const V3 ph = p - mk(Q[0]);
    a = dot(ph, mk(Q[2]));
    b = dot(ph, mk(Q[3]));
}

// The same for callers that have the ray and t: hit_quad's own arithmetic, bit for bit (a hit that came
// from a box leaf of the wide tree carries no (alpha, beta); see hit_wide_box).
__device__ __forceinline__ void quad_uv_ray(const float4 *__restrict__ Q, V3 o, V3 d, float t, float &a, float &b) {
    const V3 qo = mk(Q[0]) - o;
    const V3 ph = fma3(d, t, -qo);
    a = dot(ph, mk(Q[2]));
    b = dot(ph, mk(Q[3]));
}

// Sphere (hittable.rs:166-204).  Roots accepted in (tlo, thi) open (Interval::surrounds).
// Same quadratic, evaluated about the point of closest approach so that fp32 keeps the
// discriminant of large spheres: disc/a = r^2 - |oc - (h/a) d|^2.
// from_self: the ray starts on this sphere, so one root is 0 by construction and the other
// is 2h/a; the reference reaches the same root after rejecting |t| < 0.001.
// The quadratic a t^2 - 2 h t + c = 0 (h = d.oc, c = |oc|^2 - r^2) loses its small terms in
// fp32 twice: oc = centre - o rounds away the low bits of o when the centre is far (a ground
// sphere of radius 1000), and |oc|^2 - r^2 cancels when the origin is near the surface.  Both
// are repaired with error-free transformations (TwoSum, FMA products); the discriminant comes
// from whichever form is well conditioned: h^2 - a c near the surface, a (r^2 - |oc - (h/a) d|^2)
// far from a small sphere.  Roots are then q/a and c/q with q = h + sign(h) sqrt(disc).
__device__ __forceinline__ void two_sum(float a, float b, float &s, float &e) {
#pragma clang fp contract(off)
    s = a + b;
    float bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}
// The compensated arithmetic runs only where c = |oc|^2 - r^2 has lost more than six bits to
// cancellation: an origin within r/128 of the surface but not on it -- always for objects resting
// on a ground sphere of radius 1000, rarely otherwise, so most waves skip it.  A ray that starts
// on the sphere needs neither c nor the low part of centre - o (its root is 2h/a).
struct SphereEq { // a t^2 - 2 h t + c = 0 for one ray and one sphere, plain fp32
    V3 center, oc;
    float r, pr, a, inv_a, h, b2, disc;
};
// STATIC: the sphere does not move (s1 is not looked at)
template <bool STATIC = false>
__device__ __forceinline__ SphereEq sphere_eq(float4 s0, float4 s1, V3 o, V3 d, float time) {
#pragma clang fp contract(off) // the error terms in sphere_roots must see the same roundings as these sums
    SphereEq q;
    q.center = STATIC ? mk(s0) : fma3(mk(s1), time, mk(s0));
    q.r = s0.w, q.pr = q.r * q.r;
    q.a = len2(d);
    q.inv_a = rcp(q.a); // v_rcp_f32, 1 ulp (the compiler's 1/x costs ten instructions here)
    q.oc = q.center - o;
    q.h = dot(d, q.oc);
    q.b2 = len2(q.oc);
    // discriminant about the point of closest approach: disc/a = r^2 - |oc - (h/a) d|^2
    q.disc = q.a * fmaf(q.r, q.r, -len2(fma3(d, -(q.h * q.inv_a), q.oc)));
    return q;
}
// Both roots, t_near <= t_far; false if the line misses the sphere.
__device__ __forceinline__ bool sphere_roots(const SphereEq &q, V3 o, V3 d, float &t_near, float &t_far) {
#pragma clang fp contract(off)
    V3 oc = q.oc;
    float h = q.h, disc = q.disc;
    float c = q.b2 - q.pr;
    if (fabsf(c) < 0.015625f * fmaxf(q.b2, q.pr)) {
        V3 ol; // oc + ol = centre - o exactly
        two_sum(q.center.x, -o.x, oc.x, ol.x);
        two_sum(q.center.y, -o.y, oc.y, ol.y);
        two_sum(q.center.z, -o.z, oc.z, ol.z);
        h = dot(d, oc) + dot(d, ol);
        // c = |oc|^2 - r^2, compensated
        float px = oc.x * oc.x, py = oc.y * oc.y, pz = oc.z * oc.z;
        float low = fmaf(oc.x, oc.x, -px) + fmaf(oc.y, oc.y, -py) + fmaf(oc.z, oc.z, -pz) - fmaf(q.r, q.r, -q.pr);
        low = fmaf(2.0f, dot(oc, ol), low);
        float s1_, e1, s2_, e2, s3_, e3;
        two_sum(px, py, s1_, e1);
        two_sum(s1_, pz, s2_, e2);
        two_sum(s2_, -q.pr, s3_, e3);
        c = s3_ + (e1 + e2 + e3 + low);
        disc = fmaf(h, h, -(q.a * c)); // |c| << r^2 here: this form is the well-conditioned one
    }
    if (disc < 0.0f) return false;
    // v_sqrt_f32 and v_rcp_f32 as they are (1 ulp each): the compiler's sqrtf and / wrap them in range
    // scaling, six instructions apiece, for arguments that do not occur here
    const float w = h + copysignf(fsqrt(disc), h);
    const float ta = c * rcp(w), tb = w * q.inv_a;
    t_near = fminf(ta, tb), t_far = fmaxf(ta, tb);
    return true;
}
template <bool STATIC = false>
__device__ __forceinline__ bool hit_sphere(float4 s0, float4 s1, V3 o, V3 d, float time, float tlo, float thi,
                                           bool from_self, float &t) {
    const SphereEq q = sphere_eq<STATIC>(s0, s1, o, d, time);
    if (from_self) {
        const float root = 2.0f * q.h * q.inv_a;
        if (!(tlo < root && root < thi)) return false;
        if (q.disc < 0.0f) return false; // the reference still needs a real discriminant here
        t = root;
        return true;
    }
    float t_near, t_far;
    if (!sphere_roots(q, o, d, t_near, t_far)) return false;
    float root = t_near;
    if (!(tlo < root && root < thi)) {
        root = t_far;
        if (!(tlo < root && root < thi)) return false;
    }
    t = root;
    return true;
}
// STATIC: nothing in the scene moves -- the centre is the record's first float4, the second is not read
template <bool UNIFORM = false, bool STATIC = false>
__device__ __forceinline__ bool hit_sphere(const float4 *__restrict__ S, V3 o, V3 d, float time, float tlo,
                                           float thi, bool from_self, float &t) {
    if (STATIC) return hit_sphere<true>(load4<UNIFORM>(S, 0), make_float4(0.f, 0.f, 0.f, 0.f), o, d, 0.0f, tlo, thi, from_self, t);
    return hit_sphere(load4<UNIFORM>(S, 0), load4<UNIFORM>(S, 1), o, d, time, tlo, thi, from_self, t);
}

// hittable.rs:150-162; u_shift undoes a baked RotateY (longitude is taken in the sphere's frame)
__device__ __forceinline__ void sphere_uv(V3 n, float u_shift, float &u, float &v) {
    float theta = acosf(fminf(fmaxf(-n.y, -1.0f), 1.0f));
    float phi = atan2f(-n.z, n.x) + RT_PI;
    u = phi * (0.5f * RT_INV_PI);
    if (u_shift != 0.0f) {
        u += u_shift;
        u -= floorf(u);
    }
    v = theta * RT_INV_PI;
}

// closest boundary hit with t in (tlo, thi) for a ConstantMedium boundary list
__device__ __forceinline__ bool hit_boundary(SceneRef sc, uint32_t first, uint32_t count, V3 o, V3 d,
                                             float time, float tlo, float thi, float &tout) {
    bool any = false;
    float closest = thi;
    for (uint32_t k = 0; k < count; ++k) {
        uint32_t r = sc.refs[first + k];
        uint32_t idx = r & RT_REF_INDEX_MASK;
        float t, a, b;
        if ((r >> RT_REF_KIND_SHIFT) == RT_KIND_QUAD) {
            if (hit_quad(sc.quads + 6 * idx, o, d, tlo, closest, t, a, b)) {
                closest = t;
                any = true;
            }
        } else {
            if (hit_sphere(sc.spheres + 3 * idx, o, d, time, tlo, closest, false, t)) {
                closest = t;
                any = true;
            }
        }
    }
    tout = closest;
    return any;
}

struct RayCtx { // what a medium needs to draw its free-flight distance
    uint32_t k0, k1, pixel, sample, bounce;
};

// ConstantMedium::hit (hittable.rs:847-886)
__device__ __forceinline__ bool hit_medium(SceneRef sc, uint32_t idx, V3 o, V3 d, float time, float tlo,
                                           float thi, const RayCtx &rc, float &tout) {
    rt_medium m = sc.media[idx];
    float t1, t2;
    const uint32_t only = sc.refs[m.first];
    if (m.count == 1 && (only >> RT_REF_KIND_SHIFT) == RT_KIND_SPHERE) {
        // A boundary that is one sphere: Sphere::hit over the universe returns the near root and
        // over (t1 + 0.0001, inf) the far one (hittable.rs:181-190), so one quadratic gives both.
        const float4 *S = sc.spheres + 3 * (only & RT_REF_INDEX_MASK);
        if (!sphere_roots(sphere_eq(S[0], S[1], o, d, time), o, d, t1, t2)) return false;
        if (!(t1 + 0.0001f < t2)) return false;
    } else {
        if (!hit_boundary(sc, m.first, m.count, o, d, time, -RT_INF, RT_INF, t1)) return false;
        if (!hit_boundary(sc, m.first, m.count, o, d, time, t1 + 0.0001f, RT_INF, t2)) return false;
    }
    if (t1 < tlo) t1 = tlo;
    if (t2 > thi) t2 = thi;
    if (t1 >= t2) return false;
    if (t1 < 0.0f) t1 = 0.0f;
    float ray_len = fsqrt(len2(d));
    float dist_inside = (t2 - t1) * ray_len;
    float xi = medium_uniform(rc.k0, rc.k1, rc.pixel, rc.sample, rc.bounce, m.prim_id);
    float hit_dist = m.neg_inv_density * logf(xi);
    if (hit_dist > dist_inside) return false;
    tout = t1 + hit_dist * rcp(ray_len);
    return true;
}

// A box along the world's axes whose six faces are quads of the scene (leaf kind RT_KIND_BOX of the
// wide tree; box3, hittable.rs:398-455): the six quad tests in one.  The ray crosses the faces' planes
// where the quad tests put their hits -- t = (plane - o) * rcp(d) is hit_quad's own arithmetic for a unit
// normal along an axis -- and of a convex box only the first crossing past tlo can be the closest hit:
// the entry face, or the exit face for a ray that starts inside or on the entry face.  inv = 1/d per
// axis.  Returns the face's quad ref in `face`.  t in [tlo, thi] closed, like a quad's.
__device__ __forceinline__ bool hit_wide_box(const float4 *__restrict__ B, V3 o, V3 d, V3 inv, float tlo, float thi,
                                             uint32_t origin_ref, float &t, uint32_t &face) {
    const float4 b0 = B[0], b1 = B[1], b2 = B[2];
    const float xl = (b0.x - o.x) * inv.x, xh = (b1.x - o.x) * inv.x;
    const float yl = (b0.y - o.y) * inv.y, yh = (b1.y - o.y) * inv.y;
    const float zl = (b0.z - o.z) * inv.z, zh = (b1.z - o.z) * inv.z;
    // a ray parallel to an axis: +-inf, or NaN when it lies in a plane; min/max drop a NaN operand,
    // which leaves that plane out -- the quad test rejects such a face as parallel (|denom| < 1e-8)
    const float xn = fminf(xl, xh), xf = fmaxf(xl, xh), yn = fminf(yl, yh), yf = fmaxf(yl, yh), zn = fminf(zl, zh), zf = fmaxf(zl, zh);
    const float t_near = fmaxf(fmaxf(xn, yn), zn), t_far = fminf(fminf(xf, yf), zf);
    // the face on each axis the ray enters / leaves through follows from the direction's sign
    const bool ux = d.x > 0.0f, uy = d.y > 0.0f, uz = d.z > 0.0f;
    const uint32_t rlx = fbits(b0.w), rhx = fbits(b1.w), rly = fbits(b2.x), rlz = fbits(b2.y), rhy = fbits(b2.z), rhz = fbits(b2.w);
    const uint32_t in_x = ux ? rlx : rhx, in_y = uy ? rly : rhy, in_z = uz ? rlz : rhz;
    const uint32_t out_x = ux ? rhx : rlx, out_y = uy ? rhy : rly, out_z = uz ? rhz : rlz;
    const uint32_t near_ref = xn >= fmaxf(yn, zn) ? in_x : (yn >= zn ? in_y : in_z);
    const uint32_t far_ref = xf <= fminf(yf, zf) ? out_x : (yf <= zf ? out_y : out_z);
    const bool use_near = (t_near >= tlo) & (near_ref != origin_ref);
    const float tc = use_near ? t_near : t_far;
    const uint32_t rcand = use_near ? near_ref : far_ref;
    // (the quad test's rejection of a ray parallel to the face needs no word here: such a ray's crossings
    // of that axis are +-huge and never the first crossing of a hit)
    if (!((t_near <= t_far) & (tc >= tlo) & (tc <= thi) & (rcand != origin_ref))) return false;
    t = tc;
    face = rcand;
    return true;
}

// EXTRAS: the tree may hold box leaves (false compiles their test out: the kernels behind a tree sit at
// their register limit, and a scene without such leaves ran 3 % slower with the test compiled in).
// PLAIN: nothing moves and there is no medium (see k_paths).
template <bool UNIFORM = false, bool EXTRAS = true, bool PLAIN = false>
__device__ __forceinline__ void test_prim(SceneRef sc, uint32_t r, V3 o, V3 d, float time, float tlo,
                                          uint32_t origin_ref, const RayCtx &rc, Best &best, V3 inv = V3{0.f, 0.f, 0.f}) {
    uint32_t kind = r >> RT_REF_KIND_SHIFT, idx = r & RT_REF_INDEX_MASK;
    float t, a = 0.f, b = 0.f;
    bool h;
    if (EXTRAS && kind == RT_KIND_BOX) { // wide tree only (the caller passes inv = 1/d)
        uint32_t face = RT_REF_NONE;
        h = hit_wide_box(sc.wide_boxes + 3 * idx, o, d, inv, tlo, best.t, origin_ref, t, face);
        // between coincident faces of two boxes the later quad wins, as between two quads (below)
        if (h && t == best.t && (best.ref >> RT_REF_KIND_SHIFT) == RT_KIND_QUAD && best.ref > face) h = false;
        if (h) r = face;
        a = b = __builtin_nanf(""); // (alpha, beta) of the face: made from the hit point where a material asks (shade_emit)
    } else if (kind == RT_KIND_QUAD) {
        // a ray leaving a planar primitive cannot meet it again; the reference rejects the
        // t ~ 0 root through its 0.001 lower bound, which fp32 at scene scale cannot resolve
        if (r == origin_ref) return;
        h = hit_quad<UNIFORM>(sc.quads + 6 * idx, o, d, tlo, best.t, t, a, b);
        // Two quads met at exactly the same t (coincident faces): the reference keeps the one it
        // visits later -- HittableVec::hit and BVHNode::hit both accept t <= closest
        // (hittable.rs:252-264, boundingbox.rs:219-231) and walk left to right.  Quads are stored
        // in that visiting order, so the larger index wins whatever order a tree search takes.
        // (A sphere never takes a tie from a quad: its root must lie strictly inside the interval.)
        if (h && t == best.t && (best.ref >> RT_REF_KIND_SHIFT) == RT_KIND_QUAD && best.ref > r) h = false;
    } else if (kind == RT_KIND_SPHERE) {
        h = hit_sphere<UNIFORM, PLAIN>(sc.spheres + 3 * idx, o, d, time, tlo, best.t, r == origin_ref, t);
    } else {
        h = PLAIN ? false : hit_medium(sc, idx, o, d, time, tlo, best.t, rc, t);
    }
    if (h) {
        best.t = t;
        best.ref = r;
        best.u = a;
        best.v = b;
    }
}

// BoundingBox3::hit (boundingbox.rs:96-113) with a cached reciprocal direction.  The reference
// rejects an overlap of size 0 in f64; in fp32 the two ends of a crossing of a thin box (a quad's
// box is 1e-4 thick) can round to the same t, or cross by a few ulp when they come from
// different axes, so here the overlap may be empty by up to BOX_SLACK relative.  Entering a box
// needlessly costs time only: hits come from the primitive tests.
constexpr float BOX_SLACK = 1e-6f;
__device__ __forceinline__ bool hit_box(const float *mn, const float *mx, V3 o, V3 inv, float tlo, float thi,
                                        float &tnear) {
    float x0 = (mn[0] - o.x) * inv.x, x1 = (mx[0] - o.x) * inv.x;
    float y0 = (mn[1] - o.y) * inv.y, y1 = (mx[1] - o.y) * inv.y;
    float z0 = (mn[2] - o.z) * inv.z, z1 = (mx[2] - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tlo));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), thi));
    tnear = tn;
    return tn <= fmaf(fabsf(tf), BOX_SLACK, tf);
}

// Closest hit of the world.  `stack` is this lane's column of an LDS array [STACK_DEPTH][BLOCK];
// `lnodes` the block's LDS copy of nodes[0 .. LDS_NODES).
template <bool USE_LDS_NODES>
__device__ __forceinline__ void traverse(SceneRef sc, V3 o, V3 d, float time, float tlo, float thi,
                                         uint32_t origin_ref, const RayCtx &rc, uint32_t *stack,
                                         const rt_node *lnodes, Best &best) {
    best.t = thi;
    best.ref = RT_REF_NONE;
    best.u = best.v = 0.f;
    V3 inv = {rcp(d.x), rcp(d.y), rcp(d.z)};
    int sp = 0;
    uint32_t cur = sc.root;
    for (;;) {
        if ((cur >> RT_REF_KIND_SHIFT) == RT_KIND_NODE) {
            uint32_t idx = cur & RT_REF_INDEX_MASK;
            const rt_node *np = (USE_LDS_NODES && idx < LDS_NODES) ? (lnodes + idx) : (sc.nodes + idx);
            const float4 *n4 = reinterpret_cast<const float4 *>(np);
            float4 a = n4[0], b = n4[1], c = n4[2], e = n4[3];
            uint32_t left = fbits(a.w), right = fbits(b.w);
            if (right == RT_NODE_LEAF) {
                uint32_t count = fbits(c.w);
                for (uint32_t k = 0; k < count; ++k) {
                    uint32_t r = sc.refs[left + k];
                    if ((r >> RT_REF_KIND_SHIFT) == RT_KIND_NODE) {
                        if (sp < STACK_DEPTH) stack[(sp++) * BLOCK] = r;
                    } else {
                        test_prim(sc, r, o, d, time, tlo, origin_ref, rc, best);
                    }
                }
            } else {
                float lmn[3] = {a.x, a.y, a.z}, lmx[3] = {b.x, b.y, b.z};
                float rmn[3] = {c.x, c.y, c.z}, rmx[3] = {e.x, e.y, e.z};
                float tl, tr;
                bool hl = hit_box(lmn, lmx, o, inv, tlo, best.t, tl);
                bool hr = hit_box(rmn, rmx, o, inv, tlo, best.t, tr);
                if (hl && hr) {
                    // near child first; results do not depend on the order (closest hit)
                    bool swap = tr < tl;
                    uint32_t nearc = swap ? right : left, farc = swap ? left : right;
                    if (sp < STACK_DEPTH) stack[(sp++) * BLOCK] = farc;
                    cur = nearc;
                    continue;
                } else if (hl) {
                    cur = left;
                    continue;
                } else if (hr) {
                    cur = right;
                    continue;
                }
            }
        } else {
            test_prim(sc, cur, o, d, time, tlo, origin_ref, rc, best);
        }
        if (sp == 0) break;
        cur = stack[(--sp) * BLOCK];
    }
}

// Closest hit through the 4-wide BVH.  `stack` is this lane's column of an LDS array
// [depth][BLOCK] whose depth the host sized to the scene's exact need (rt_accel.h).
// Two nested loops: the inner one walks interior nodes only, the primitive test runs once all
// lanes of the wave have left it, so a wave does not execute node code and primitive code for
// one lane each per iteration.  Children are visited nearest first; the others go on the stack
// farthest first.  Visiting order cannot change the closest hit.
// The top of the tree (nodes are numbered breadth first) is staged in LDS by stage_nodes4: every
// ray starts there, and a per-lane 16-byte read costs the LDS a fraction of what 64 scattered
// cache lines cost the vector memory path.  A lane reads a node through a generic pointer that
// points into LDS or into global memory, so both cases are one instruction stream.
template <bool QUANT> __device__ __forceinline__ void stage_nodes4(SceneRef sc, uint32_t *s_dyn) {
    const uint32_t n = sc.lds_nodes4 * (QUANT ? 4u : 8u);
    float4 *dst = reinterpret_cast<float4 *>(s_dyn + size_t(sc.stack4) * BLOCK);
    for (uint32_t i = threadIdx.x; i < n; i += BLOCK) dst[i] = sc.nodes4[i];
    __syncthreads();
}
// Wide traversal, resumable.  A lane's traversal is (cur, sp, best): cur is the node or primitive
// it handles next (RT_REF_NONE: none -- finished, or not traversing), sp its LDS stack height.
// One round = node steps while enough lanes are at nodes, then one primitive test for the lanes
// that hold one.  Lanes with cur == RT_REF_NONE take no part.
// STAGED = false reads every node from global memory with plain global loads (generic loads
// cost ~13 % on a scene too deep to leave LDS for nodes).
__device__ __forceinline__ void trav4_begin(float thi, uint32_t &cur, uint32_t &sp, Best &best) {
    best.t = thi;
    best.ref = RT_REF_NONE;
    best.u = best.v = 0.f;
    best.cls = RT4_CLASS_END; // a miss ends the path like a light
    sp = 0;
    cur = RT_KIND_NODE << RT_REF_KIND_SHIFT; // node 0
}
// Where a lane's traversal keeps its stack and finds staged nodes.  The stack is a column of an LDS
// array (entry i at stack[i * stride]).  SPILL: only the first lds_entries entries live there, deeper
// ones in global memory (entry i at spill[(i - lds_entries) * spill_stride]) -- for kernels that
// need the LDS for something else; pushes beyond a handful of pending siblings are rare.
// The two parts of a stack are named by address space, so that a push is an LDS store (and, behind
// a rarely taken branch, a global store) instead of one generic store through a selected pointer.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) uint32_t global_u32;
struct TravMem {
    lds_u32 *stack;       // this lane's column: entry i at stack[i * STRIDE]
    const char *l_nodes;  // LDS copy of the first nodes
    uint32_t lds_bytes;   // of l_nodes
    uint32_t lds_entries; // SPILL only
    global_u32 *spill;    // SPILL only: this lane's column
    uint32_t spill_stride;
};
__device__ __forceinline__ TravMem block_trav_mem(SceneRef sc, uint32_t *s_dyn, uint32_t node_bytes) {
    // kernels whose whole dynamic LDS is [stack4][BLOCK] stack words followed by the staged nodes
    return TravMem{(lds_u32 *)(s_dyn + threadIdx.x), reinterpret_cast<const char *>(s_dyn + size_t(sc.stack4) * BLOCK),
                   sc.lds_nodes4 * node_bytes, 0u, nullptr, 0u};
}
template <bool SPILL, uint32_t STRIDE> __device__ __forceinline__ void trav_push(const TravMem &m, uint32_t &sp, uint32_t v) {
    if (SPILL && __builtin_expect(sp >= m.lds_entries, 0)) m.spill[size_t(sp - m.lds_entries) * m.spill_stride] = v;
    else m.stack[sp * STRIDE] = v;
    ++sp;
}
template <bool SPILL, uint32_t STRIDE> __device__ __forceinline__ uint32_t trav_pop(const TravMem &m, uint32_t &sp) {
    --sp;
    if (SPILL && __builtin_expect(sp >= m.lds_entries, 0)) return m.spill[size_t(sp - m.lds_entries) * m.spill_stride];
    return m.stack[sp * STRIDE];
}
// QUANT: 64-byte nodes.  The child boxes are bytes on a grid of the node's own -- plane = origin +
// q * 2^e per axis, lower planes rounded down and upper planes up by the builder (rt_accel.h), so a
// quantised box contains the fp32 one and the closest hit is unchanged; only a few more boxes are
// entered.  A visit is four 16-byte loads per lane instead of seven and half the bytes: on a large
// scene the vector memory path, not arithmetic, is what a traversal step costs.
//   dwords 0-2 origin, 3 = exponents (biased, one byte per axis), 4-6 lower planes x y z (one byte
//   per child), 7-9 upper planes, 10-11 unused, 12-15 child refs
// t = (origin + q s - o) / d is evaluated as q * (s / d) + (origin - o) / d: two constants per axis
// and node, then one conversion and one fma per plane.
// One visit of a 4-wide node: its four child boxes against the ray, the children that are met
// sorted nearest first.  key[i] = the entry distance's bits (tn >= tlo > 0: float order = unsigned
// order), 0xFFFFFFFF for a child that is missed or absent; ref[i] = the child's word.
struct RayConsts { // per ray, hoisted out of a traversal loop by the caller
    float ox, oy, oz, ix, iy, iz;
    uint32_t nx, ny, nz, fx, fy, fz; // fp32 nodes: byte offsets of the near / far planes by the direction's signs
    bool dnx, dny, dnz;              // byte-box nodes: the ray travels downwards on that axis
};
__device__ __forceinline__ RayConsts ray_consts(V3 o, V3 d) {
    RayConsts c;
    c.ox = o.x, c.oy = o.y, c.oz = o.z;
    c.ix = rcp(d.x), c.iy = rcp(d.y), c.iz = rcp(d.z);
    // The plane a ray meets first on each axis follows from the sign of its direction, so the
    // near and far planes are chosen by the load address (lo at float4 0..2, hi at 3..5) and no
    // per-axis min/max is needed.  Sign bit, not `< 0`: -0 has reciprocal -inf.
    c.dnx = (fbits(d.x) >> 31) != 0u, c.dny = (fbits(d.y) >> 31) != 0u, c.dnz = (fbits(d.z) >> 31) != 0u;
    c.nx = (fbits(d.x) >> 31) * 48u, c.ny = 16u + (fbits(d.y) >> 31) * 48u, c.nz = 32u + (fbits(d.z) >> 31) * 48u;
    c.fx = 48u - c.nx, c.fy = 80u - c.ny, c.fz = 112u - c.nz;
    return c;
}
template <bool STAGED, bool QUANT, bool SORT>
__device__ __forceinline__ void wide_node_visit(SceneRef sc, const char *l_nodes, uint32_t lds_bytes, const RayConsts &rcn, float tlo,
                                                float best_t, uint32_t cur, uint32_t (&key)[4], uint32_t (&ref)[4]) {
    typedef float v2f __attribute__((ext_vector_type(2))); // two children per packed instruction
    constexpr uint32_t NODE_BYTES = QUANT ? 64u : 128u;
    const char *g_nodes = reinterpret_cast<const char *>(sc.nodes4);
    const v2f ox = {rcn.ox, rcn.ox}, oy = {rcn.oy, rcn.oy}, oz = {rcn.oz, rcn.oz};
    const v2f ix = {rcn.ix, rcn.ix}, iy = {rcn.iy, rcn.iy}, iz = {rcn.iz, rcn.iz};
    const uint32_t nx = rcn.nx, ny = rcn.ny, nz = rcn.nz, fx = rcn.fx, fy = rcn.fy, fz = rcn.fz;
    const uint32_t nb = (cur & RT_REF_INDEX_MASK) * NODE_BYTES; // the wide array stays below 4 GiB (checked at upload)
    const char *node = (STAGED && nb < lds_bytes ? l_nodes : g_nodes) + nb;
    auto ld = [&](uint32_t byte_off) { return *reinterpret_cast<const float4 *>(node + byte_off); };
    v2f nx01, nx23, ny01, ny23, nz01, nz23, fx01, fx23, fy01, fy23, fz01, fz23;
    if constexpr (QUANT) {
        const float4 h0 = ld(0u), p0 = ld(16u), p1 = ld(32u), cf = ld(48u);
        ref[0] = fbits(cf.x), ref[1] = fbits(cf.y), ref[2] = fbits(cf.z), ref[3] = fbits(cf.w);
        const uint32_t ex = fbits(h0.w);
        const float sx = bitsf((ex & 0xFFu) << 23), sy = bitsf(((ex >> 8) & 0xFFu) << 23), sz = bitsf(((ex >> 16) & 0xFFu) << 23);
        const v2f ax = {sx * ix.x, sx * ix.x}, ay = {sy * iy.x, sy * iy.x}, az = {sz * iz.x, sz * iz.x};
        const float bxs = (h0.x - rcn.ox) * ix.x, bys = (h0.y - rcn.oy) * iy.x, bzs = (h0.z - rcn.oz) * iz.x;
        const v2f bx = {bxs, bxs}, by = {bys, bys}, bz = {bzs, bzs};
        // the plane a ray meets first on an axis is the lower one unless it travels downwards
        const uint32_t lox = fbits(p0.x), loy = fbits(p0.y), loz = fbits(p0.z), hix = fbits(p0.w), hiy = fbits(p1.x), hiz = fbits(p1.y);
        const bool dnx = rcn.dnx, dny = rcn.dny, dnz = rcn.dnz;
        const uint32_t qnx = dnx ? hix : lox, qfx = dnx ? lox : hix;
        const uint32_t qny = dny ? hiy : loy, qfy = dny ? loy : hiy;
        const uint32_t qnz = dnz ? hiz : loz, qfz = dnz ? loz : hiz;
        auto b01 = [](uint32_t w) { return v2f{float(w & 0xFFu), float((w >> 8) & 0xFFu)}; };
        auto b23 = [](uint32_t w) { return v2f{float((w >> 16) & 0xFFu), float(w >> 24)}; };
        // 0 * inf = NaN for a ray parallel to an axis: max3/min3 drop NaN operands, which leaves
        // that axis unconstrained -- a box entered needlessly, never a box missed
        nx01 = b01(qnx) * ax + bx, nx23 = b23(qnx) * ax + bx, fx01 = b01(qfx) * ax + bx, fx23 = b23(qfx) * ax + bx;
        ny01 = b01(qny) * ay + by, ny23 = b23(qny) * ay + by, fy01 = b01(qfy) * ay + by, fy23 = b23(qfy) * ay + by;
        nz01 = b01(qnz) * az + bz, nz23 = b23(qnz) * az + bz, fz01 = b01(qfz) * az + bz, fz23 = b23(qfz) * az + bz;
    } else {
        const float4 ax = ld(nx), ay = ld(ny), az = ld(nz);
        const float4 bx = ld(fx), by = ld(fy), bz = ld(fz);
        const float4 cf = ld(96u);
        ref[0] = fbits(cf.x), ref[1] = fbits(cf.y), ref[2] = fbits(cf.z), ref[3] = fbits(cf.w);
        // BoundingBox3::hit (boundingbox.rs:96-113) for four boxes; see hit_box for the slack.
        // 0 * inf = NaN where the origin lies on a plane the ray is parallel to: max3/min3
        // drop NaN operands, which leaves that axis unconstrained.
        nx01 = (v2f{ax.x, ax.y} - ox) * ix, nx23 = (v2f{ax.z, ax.w} - ox) * ix;
        ny01 = (v2f{ay.x, ay.y} - oy) * iy, ny23 = (v2f{ay.z, ay.w} - oy) * iy;
        nz01 = (v2f{az.x, az.y} - oz) * iz, nz23 = (v2f{az.z, az.w} - oz) * iz;
        fx01 = (v2f{bx.x, bx.y} - ox) * ix, fx23 = (v2f{bx.z, bx.w} - ox) * ix;
        fy01 = (v2f{by.x, by.y} - oy) * iy, fy23 = (v2f{by.z, by.w} - oy) * iy;
        fz01 = (v2f{bz.x, bz.y} - oz) * iz, fz23 = (v2f{bz.z, bz.w} - oz) * iz;
    }
    const float tn[4] = {fmaxf(fmaxf(fmaxf(nx01.x, ny01.x), nz01.x), tlo), fmaxf(fmaxf(fmaxf(nx01.y, ny01.y), nz01.y), tlo),
                         fmaxf(fmaxf(fmaxf(nx23.x, ny23.x), nz23.x), tlo), fmaxf(fmaxf(fmaxf(nx23.y, ny23.y), nz23.y), tlo)};
    const float tf[4] = {fminf(fminf(fminf(fx01.x, fy01.x), fz01.x), best_t), fminf(fminf(fminf(fx01.y, fy01.y), fz01.y), best_t),
                         fminf(fminf(fminf(fx23.x, fy23.x), fz23.x), best_t), fminf(fminf(fminf(fx23.y, fy23.y), fz23.y), best_t)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool hit = (tn[i] <= fmaf(fabsf(tf[i]), BOX_SLACK, tf[i])) & (ref[i] != RT_REF_NONE);
        key[i] = hit ? fbits(tn[i]) : 0xFFFFFFFFu; // tn >= tlo > 0: float order = unsigned order
    }
    // SORT: the four (key, ref) pairs ascending, 5 compare-exchanges
    if constexpr (SORT) {
#define RT_CSWAP(a, b)                                                                       \
{                                                                                        \
const bool sw = key[b] < key[a];                                                     \
const uint32_t ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];                 \
const uint32_t ra_ = sw ? ref[b] : ref[a], rb_ = sw ? ref[a] : ref[b];               \
key[a] = ka, key[b] = kb, ref[a] = ra_, ref[b] = rb_;                                \
}
    RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
    }
}

// Rounds for the lanes that call it together, until `more(lanes still traversing, lanes finished)`
// says no (asked after every round, wave-uniform; both counts are of the calling lanes).  The
// per-ray constants are made once per call: a caller that came back for every round paid for them
// -- three reciprocals and the plane offsets -- once per primitive test.  Returns the rounds run.
template <bool STAGED, bool QUANT = false, bool SPILL = false, uint32_t STRIDE = BLOCK, bool EXTRAS = true, bool PLAIN = false, typename More>
__device__ __forceinline__ uint32_t trav4_rounds(SceneRef sc, V3 o, V3 d, float time, float tlo, uint32_t origin_ref,
                                                 const RayCtx &rc, const TravMem &mem, uint32_t &cur, uint32_t &sp, Best &best,
                                                 More more, TravStats *ts = nullptr) {
    const RayConsts rcn = ray_consts(o, d);
    uint32_t rounds = 0;
    do {
        ++rounds;
        // Node steps run while the lanes at a node outnumber the lanes that hold a primitive by
        // NODE_SHARE to one; then the lanes that hold a primitive test it while the node lanes wait.
        // Waiting for the last node lane instead idles most of the wave.  (A fixed quorum of 16 node
        // lanes, the rule before this one, was 4 % slower on the Book-2 final scene and 6 % slower on
        // the 1 M-primitive scene: what matters is how many lanes each kind of step would leave idle.)
        for (;;) {
            const bool at_node = (cur >> RT_REF_KIND_SHIFT) == RT_KIND_NODE;
            const uint32_t n_node = uint32_t(__popcll(__ballot(at_node)));
            if (n_node == 0) break;
            if (n_node < NODE_SHARE * uint32_t(__popcll(__ballot(!at_node && cur != RT_REF_NONE)))) break;
#ifdef RT_REGROUP_STATS
            if (ts) ts->node_steps += 1, ts->node_lanes += n_node;
#endif
            if (!at_node) continue;
            uint32_t key[4], ref[4];
            // Large trees (the ones that take byte boxes) are searched with all four children in
            // order of distance; on small trees, which live in LDS and whose cost is instructions, only
            // the nearest child is singled out (measured: +3 to +5 % on trees of hundreds of nodes,
            // -4 % at 466 k nodes, where every needless visit is a trip to memory).
    #ifdef RT_SORT_POLICY
            constexpr bool SORT = RT_SORT_POLICY == 2 || (RT_SORT_POLICY == 1 && QUANT);
    #else
            constexpr bool SORT = QUANT;
    #endif
            wide_node_visit<STAGED, QUANT, SORT>(sc, mem.l_nodes, mem.lds_bytes, rcn, tlo, best.t, cur, key, ref);
            if constexpr (SORT) {
                if (key[3] != 0xFFFFFFFFu) trav_push<SPILL, STRIDE>(mem, sp, ref[3]);
                if (key[2] != 0xFFFFFFFFu) trav_push<SPILL, STRIDE>(mem, sp, ref[2]);
                if (key[1] != 0xFFFFFFFFu) trav_push<SPILL, STRIDE>(mem, sp, ref[1]);
                if (key[0] != 0xFFFFFFFFu) {
                    cur = ref[0];
                } else if (sp) {
                    cur = trav_pop<SPILL, STRIDE>(mem, sp);
                } else {
                    cur = RT_REF_NONE;
                }
            } else {
                // The nearest child is visited next, the other children that are met go on the stack as
                // they stand in the node: a full sort of the four costs a quarter of a visit's
                // instructions, and the order of the far children only decides how soon they are culled.
                const uint32_t nearest = min(min(key[0], key[1]), min(key[2], key[3]));
                const bool n0 = key[0] == nearest, n1 = !n0 & (key[1] == nearest), n2 = !(n0 | n1) & (key[2] == nearest);
                const bool n3 = !(n0 | n1 | n2);
                if ((key[3] != 0xFFFFFFFFu) & !n3) trav_push<SPILL, STRIDE>(mem, sp, ref[3]);
                if ((key[2] != 0xFFFFFFFFu) & !n2) trav_push<SPILL, STRIDE>(mem, sp, ref[2]);
                if ((key[1] != 0xFFFFFFFFu) & !n1) trav_push<SPILL, STRIDE>(mem, sp, ref[1]);
                if ((key[0] != 0xFFFFFFFFu) & !n0) trav_push<SPILL, STRIDE>(mem, sp, ref[0]);
                if (nearest != 0xFFFFFFFFu) {
                    cur = n0 ? ref[0] : (n1 ? ref[1] : (n2 ? ref[2] : ref[3]));
                } else if (sp) {
                    cur = trav_pop<SPILL, STRIDE>(mem, sp);
                } else {
                    cur = RT_REF_NONE;
                }
            }
        }
#ifdef RT_REGROUP_STATS
        if (ts) {
            const uint32_t n_prim = uint32_t(__popcll(__ballot(cur != RT_REF_NONE && (cur >> RT_REF_KIND_SHIFT) != RT_KIND_NODE)));
            ts->prim_steps += n_prim != 0, ts->prim_lanes += n_prim;
        }
#endif
        if (cur != RT_REF_NONE && (cur >> RT_REF_KIND_SHIFT) != RT_KIND_NODE) {
            // a leaf's word carries the class of its material above the index (rt_accel.h)
            const uint32_t before = best.ref;
            test_prim<false, EXTRAS, PLAIN>(sc, cur & ~RT4_CLASS_MASK, o, d, time, tlo, origin_ref, rc, best, V3{rcn.ix, rcn.iy, rcn.iz});
            if (best.ref != before) best.cls = (cur & RT4_CLASS_MASK) >> RT4_CLASS_SHIFT;
            cur = sp ? trav_pop<SPILL, STRIDE>(mem, sp) : RT_REF_NONE;
        }
    } while (more(uint32_t(__popcll(__ballot(cur != RT_REF_NONE))), uint32_t(__popcll(__ballot(cur == RT_REF_NONE)))));
    return rounds;
}

// The primitives the upload kept out of the wide tree because they enclose the rest of the scene
// (DScene::post_ref): tested against the closest hit the walk found, by all lanes whose walk has ended
// at once -- inside the walk each lane would get to them at a moment of its own.  The closest hit does
// not depend on the order of the tests (a medium's draw is keyed by the ray, not by the order).
__device__ __forceinline__ void test_post(SceneRef sc, V3 o, V3 d, float time, float tlo, uint32_t origin_ref, const RayCtx &rc, Best &best) {
    const uint32_t n = sc.n_post; // uniform
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t word = sc.post_ref[i], before = best.ref;
        test_prim<true>(sc, word & ~RT4_CLASS_MASK, o, d, time, tlo, origin_ref, rc, best);
        if (best.ref != before) best.cls = (word & RT4_CLASS_MASK) >> RT4_CLASS_SHIFT;
    }
}

template <bool STAGED, bool QUANT>
__device__ __forceinline__ void traverse4(SceneRef sc, V3 o, V3 d, float time, float tlo, float thi,
                                          uint32_t origin_ref, const RayCtx &rc, uint32_t *s_dyn, Best &best) {
    uint32_t cur, sp;
    trav4_begin(thi, cur, sp, best);
    const TravMem mem = block_trav_mem(sc, s_dyn, QUANT ? 64u : 128u);
    // until every lane of the wave is done
    trav4_rounds<STAGED, QUANT>(sc, o, d, time, tlo, origin_ref, rc, mem, cur, sp, best, [](uint32_t n_trav, uint32_t) { return n_trav != 0; });
    test_post(sc, o, d, time, tlo, origin_ref, rc, best);
}

// One entry for the three structures.  s_stack / s_nodes: static LDS of the binary traversal;
// s_dyn: dynamic LDS of the wide traversal.
template <int ACCEL, bool USE_LDS_NODES>
__device__ __forceinline__ void closest_hit(SceneRef sc, V3 o, V3 d, float time, float tlo, float thi, uint32_t origin_ref,
                                            const RayCtx &rc, uint32_t *s_stack, const rt_node *s_nodes, uint32_t *s_dyn,
                                            Best &best);

// Flat world (a HittableVec of primitives, e.g. main.rs:30-69): every lane walks the same list,
// so each primitive is one scalar record applied to 64 rays.  The quad test is written without
// branches: in a 64-lane wave some lane almost always passes the plane test, so skipping the
// interior test saves nothing and costs exec-mask bookkeeping.  The next record is requested
// before the current one is used; scalar loads return out of order behind one counter, so the
// wait at the end of an iteration is the only one.
// Quads are scanned first, then the rest, each group in list order.  The outcome equals one pass
// in list order: quads accept t <= closest (a later quad wins a tie), spheres need t < closest
// (an earlier hit wins), so a quad wins a quad/sphere tie in either order.
struct QuadRec {
    float f[16];
};
__device__ __forceinline__ QuadRec load_quad_rec(const RT_CONST_AS float *p) {
    QuadRec r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.f[i] = p[i];
    return r;
}
// BOXED: the caller knows that the list's quads are all rectangle or box records and that whatever else
// it holds is a packed sphere (a Cornell box): the loops over pairs, general quads, boxes in a general
// frame and primitives by reference are compiled out -- with them in, their registers and their
// hoisted set-up are paid for by a scan that never runs them.
template <bool BOXED = false>
__device__ __forceinline__ void scan_flat(SceneRef sc, V3 o, V3 d, float time, float tlo, float thi,
                                          uint32_t origin_ref, const RayCtx &rc, Best &best) {
    best.t = thi;
    best.ref = RT_REF_NONE;
    best.u = best.v = 0.f;
    const RT_CONST_AS float *Q = as_uniform(sc.flat_quads);
    const uint32_t nq = sc.n_flat_quads;
    auto test = [&](const QuadRec &rec) { // hittable.rs:341-371
        const float *c = rec.f;
        const uint32_t r = fbits(c[3]);
        const V3 n = {c[4], c[5], c[6]};
        const float denom = dot(n, d);
        const V3 qo = {c[0] - o.x, c[1] - o.y, c[2] - o.z};
        const float tt = dot(n, qo) * __builtin_amdgcn_rcpf(denom); // v_rcp_f32: 1 ulp
        const V3 ph = fma3(d, tt, -qo);
        const float alpha = dot(ph, V3{c[8], c[9], c[10]});
        const float beta = dot(ph, V3{c[12], c[13], c[14]});
        const uint32_t shape = fbits(c[11]);
        // interior rules (hittable.rs:360, 539, 660); sqrt(x) > 0.5 <=> x > 0.25 exactly.
        // One measure per shape, selected by the uniform shape code: no branches in the loop.
        const float ac = 0.5f - alpha, bc = 0.5f - beta;
        const float m_par = fmaxf(alpha, beta), m_tri = alpha + beta, m_disc = fmaf(ac, ac, bc * bc);
        const float measure = shape == RT_SHAPE_PARALLELOGRAM ? m_par : (shape == RT_SHAPE_TRIANGLE ? m_tri : m_disc);
        const float limit = shape == RT_SHAPE_DISC ? 0.25f : 1.0f;
        const bool inside = (fminf(alpha, beta) >= 0.0f) & !(measure > limit);
        const bool ok = inside & (fabsf(denom) >= 1e-8f) & (tt >= tlo) & (tt <= best.t) & (r != origin_ref);
        best.t = ok ? tt : best.t;
        best.ref = ok ? r : best.ref;
    };
    // Two records in turn, so that no register copies sit between a request and its wait.  The
    // packed array ends with an all-zero record after an odd count (a zero normal never hits)
    // and one more after that, so requests need no bounds and the loop body has no branch.
    // The asm statements keep each request where it is written.
    if (!BOXED && sc.n_flat_pairs) {
        // Parallelograms only (the usual case: walls and boxes): two quads per record, their
        // fields interleaved so that every packed instruction takes its scalar operand straight
        // from the loaded register pair, and the accept rule folded into two max3 and a compare
        // -- 45 vector and ~10 scalar instructions per pair (the general loop below: ~60 and ~75;
        // the scalar unit issues as slowly as the vector units, so both count).
        //   pair record, 32 floats: qx qy qz nx ny nz ax ay az bx by bz (2 each), ref[2], mat[2], pad
        typedef float v2f __attribute__((ext_vector_type(2)));
        const RT_CONST_AS float *P = as_uniform(sc.flat_pairs);
        const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
        const v2f dx = {d.x, d.x}, dy = {d.y, d.y}, dz = {d.z, d.z};
        for (uint32_t k = 0; k < sc.n_flat_pairs; ++k, P += 32) {
            auto f2 = [&](int i) { return v2f{P[2 * i], P[2 * i + 1]}; };
            // the whole record is requested at once, behind one wait (left alone, the compiler
            // fetches refs and materials only where a hit needs them, each behind its own wait)
            const float ref_bits[2] = {P[24], P[25]};
            const v2f first = f2(0), last = f2(11);
            asm volatile("" ::"s"(ref_bits[0]), "s"(ref_bits[1]), "s"(first), "s"(last));
            const v2f nx = f2(3), ny = f2(4), nz = f2(5);
            const v2f denom = nx * dx + ny * dy + nz * dz;
            const v2f qox = f2(0) - ox, qoy = f2(1) - oy, qoz = f2(2) - oz;
            const v2f num = nx * qox + ny * qoy + nz * qoz;
            const v2f tt = num * v2f{__builtin_amdgcn_rcpf(denom.x), __builtin_amdgcn_rcpf(denom.y)};
            const v2f phx = dx * tt - qox, phy = dy * tt - qoy, phz = dz * tt - qoz;
            const v2f alpha = phx * f2(6) + phy * f2(7) + phz * f2(8);
            const v2f beta = phx * f2(9) + phy * f2(10) + phz * f2(11);
#pragma unroll
            for (int j = 0; j < 2; ++j) { // in list order: the second sees the first's closest
                const float a = alpha[j], b = beta[j], t = tt[j];
                const uint32_t r = fbits(ref_bits[j]);
                // 0 <= a, b <= 1 and tlo <= t <= closest  <=>  the largest violation is <= 0
                // (differences keep their sign exactly; NaN only arises from denom = 0, rejected below)
                const float outside = fmaxf(fmaxf(-a, -b), fmaxf(a, b) - 1.0f);
                const float worst = fmaxf(fmaxf(outside, tlo - t), t - best.t);
                const bool ok = (worst <= 0.0f) & (fabsf(denom[j]) >= 1e-8f) & (r != origin_ref);
                best.t = ok ? t : best.t;
                best.ref = ok ? r : best.ref;
            }
        }
    } else if (!BOXED && nq) {
        QuadRec a = load_quad_rec(Q);
        for (uint32_t k = 0; k < nq; k += 2) {
            const QuadRec b = load_quad_rec(Q + 16 * size_t(k + 1));
            asm volatile("" ::: "memory");
            test(a);
            a = load_quad_rec(Q + 16 * size_t(k + 2));
            asm volatile("" ::: "memory");
            test(b);
        }
    }
    // Boxes: six parallelograms that are the faces of one rectangular box cost one slab test in
    // the box's frame instead of three pairs (135 vector instructions).  The ray crosses the
    // faces' planes where the six quad tests put their hits -- t = (plane - e.o) / (e.d) with the
    // face normal +-e -- and of a convex box only the first crossing past tlo can be the closest
    // hit: the entry face, or the exit face for a ray that starts inside or on the entry face.
    //   record, 32 floats: frame e0 e1 e2, centre c, planes (lo, hi) per axis, refs of the lo
    //   faces [3] and of the hi faces [3], class
    // The class says how much of the frame is the world's own (the upload brings the frame into
    // that form): every rigid map the reference has is a RotateY and a Translate
    // (hittable.rs:760-1001), so a box's second axis is the world's y, and the walls of a room are
    // along all three.  Such an axis needs no dot products, its planes are world coordinates, and
    // its crossing (plane - o) * rcp(d) is the quad test's own arithmetic for a face with a unit
    // normal, bit for bit.  General frame ~70 vector instructions, upright ~50, world-aligned ~40.
    typedef float v2f __attribute__((ext_vector_type(2)));
    struct BoxAxis {
        float t_lo, t_hi;
        uint32_t r_in, r_out;
    };
    // The records come sorted by class (aligned, upright, general), one loop per class: a class
    // word read per box would put a second scalar-load round trip into every iteration.
    const RT_CONST_AS float *B = as_uniform(sc.flat_boxes);
    const uint32_t n_boxes = sc.n_flat_boxes, n_aligned = sc.n_flat_boxes_aligned, n_upright = sc.n_flat_boxes_upright;
    const uint32_t n_rects = sc.n_flat_rects, n_rects_x = sc.n_flat_rects_x, n_rects_y = sc.n_flat_rects_y;
    asm volatile("" ::"s"(B), "s"(n_boxes), "s"(n_aligned), "s"(n_upright), "s"(n_rects), "s"(n_rects_x), "s"(n_rects_y)); // one request, one wait
    // Rectangles along the world's axes (a lamp under a ceiling): the quad test with the zeros of
    // its record left out -- t = (plane - o_n) * rcp(d_n), alpha and beta one coordinate of the hit
    // point each -- which gives the packed quad test's bits, so ties fall as they did.  Records of 8
    // floats ahead of the box records: plane, Q_a, Q_b, 1/extent_a, 1/extent_b, ref, -, normal axis
    // (a < b are the two other axes); sorted by normal axis, list order within one (rectangles in
    // one plane have one normal axis), one loop per axis.  ~17 vector instructions instead of 45 a pair.
    if (n_rects) {
        auto rect = [&](float on, float dn, float oa, float da, float ob, float db) {
            // (t is rounded here, as the quad test's is: contracted into the comparisons below, the
            // unrounded product would tell it from an equal t of the same plane and lose the tie)
            float t = (B[0] - on) * __builtin_amdgcn_rcpf(dn);
            asm volatile("" : "+v"(t));
            const float a = fmaf(da, t, oa - B[1]) * B[3], b = fmaf(db, t, ob - B[2]) * B[4];
            const uint32_t r = fbits(B[5]);
            const float outside = fmaxf(fmaxf(-a, -b), fmaxf(a, b) - 1.0f);
            const float worst = fmaxf(fmaxf(outside, tlo - t), t - best.t);
            const bool ok = (worst <= 0.0f) & (fabsf(dn) >= 1e-8f) & (r != origin_ref);
            best.t = ok ? t : best.t;
            best.ref = ok ? r : best.ref;
        };
        uint32_t k = 0;
        for (; k < n_rects_x; ++k, B += 8) rect(o.x, d.x, o.y, d.y, o.z, d.z);
        for (const uint32_t end = k + n_rects_y; k < end; ++k, B += 8) rect(o.y, d.y, o.x, d.x, o.z, d.z);
        for (; k < n_rects; ++k, B += 8) rect(o.z, d.z, o.x, d.x, o.y, d.y);
        B += 8 * ((0u - n_rects) & 3u); // the box records start on a 128-byte boundary
    }
    if (n_boxes) {
        // planes: (lo, hi) along the axis, od: the origin there, dd: the direction there
        auto axis = [&](int a, float od, float dd, float inv) {
            const v2f planes = {B[12 + 2 * a], B[13 + 2 * a]};
            const v2f t2 = (planes - v2f{od, od}) * v2f{inv, inv};
            BoxAxis r;
            r.t_lo = fminf(t2.x, t2.y), r.t_hi = fmaxf(t2.x, t2.y);
            const bool up = dd > 0.0f; // enters through the lo face, leaves through the hi face
            const uint32_t rl = fbits(B[18 + a]), rh = fbits(B[21 + a]);
            r.r_in = up ? rl : rh, r.r_out = up ? rh : rl;
            return r;
        };
        auto closest = [&](const BoxAxis &x, const BoxAxis &y, const BoxAxis &z) {
            const float t_near = fmaxf(fmaxf(x.t_lo, y.t_lo), z.t_lo), t_far = fminf(fminf(x.t_hi, y.t_hi), z.t_hi);
            const uint32_t near_ref = x.t_lo >= fmaxf(y.t_lo, z.t_lo) ? x.r_in : (y.t_lo >= z.t_lo ? y.r_in : z.r_in);
            const uint32_t far_ref = x.t_hi <= fminf(y.t_hi, z.t_hi) ? x.r_out : (y.t_hi <= z.t_hi ? y.r_out : z.r_out);
            // (a face may be absent -- a room open on one side: RT_REF_NONE, never a hit)
            const bool use_near = (t_near >= tlo) & (near_ref != origin_ref) & (near_ref != RT_REF_NONE);
            const float tc = use_near ? t_near : t_far;
            const uint32_t rcand = use_near ? near_ref : far_ref;
            // A later quad wins a tie (t <= closest, hittable.rs:252-264), and ties happen: a box
            // stands on the floor, its bottom face in the floor's plane.  The quad records give such
            // a pair the same t to the bit; the slab arithmetic of a general frame differs from
            // theirs in the last bits, so the tie is taken within 4 ulp.
            const bool ok = (t_near <= t_far) & (tc >= tlo) & (tc <= best.t * (1.0f + 0x1p-21f)) & (rcand != origin_ref) & (rcand != RT_REF_NONE);
            best.t = ok ? tc : best.t;
            best.ref = ok ? rcand : best.ref;
        };
        const float inv_dy = __builtin_amdgcn_rcpf(d.y);
        uint32_t k = 0;
        if (n_aligned) {
            const float inv_dx = __builtin_amdgcn_rcpf(d.x), inv_dz = __builtin_amdgcn_rcpf(d.z);
            for (; k < n_aligned; ++k, B += 32) closest(axis(0, o.x, d.x, inv_dx), axis(1, o.y, d.y, inv_dy), axis(2, o.z, d.z, inv_dz));
        }
        for (const uint32_t end = k + n_upright; k < end; ++k, B += 32) {
            // e0 and e2 lie in the xz plane; their planes are measured from the centre
            const float ocx = o.x - B[9], ocz = o.z - B[11];
            const float od0 = fmaf(B[0], ocx, B[2] * ocz), dd0 = fmaf(B[0], d.x, B[2] * d.z);
            const float od2 = fmaf(B[6], ocx, B[8] * ocz), dd2 = fmaf(B[6], d.x, B[8] * d.z);
            closest(axis(0, od0, dd0, __builtin_amdgcn_rcpf(dd0)), axis(1, o.y, d.y, inv_dy), axis(2, od2, dd2, __builtin_amdgcn_rcpf(dd2)));
        }
        for (; !BOXED && k < n_boxes; ++k, B += 32) {
            const V3 oc = {o.x - B[9], o.y - B[10], o.z - B[11]};
            float od[3], dd[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const V3 e = {B[3 * a], B[3 * a + 1], B[3 * a + 2]};
                od[a] = dot(e, oc), dd[a] = dot(e, d);
            }
            closest(axis(0, od[0], dd[0], __builtin_amdgcn_rcpf(dd[0])), axis(1, od[1], dd[1], __builtin_amdgcn_rcpf(dd[1])),
                    axis(2, od[2], dd[2], __builtin_amdgcn_rcpf(dd[2])));
        }
    }
    // the rest of the list: spheres by their packed records when there is nothing else, or by reference
    const RT_CONST_AS float *S = as_uniform(sc.flat_spheres);
    for (uint32_t k = 0; k < sc.n_flat_spheres; ++k, S += 16) {
        const float4 s0 = make_float4(S[0], S[1], S[2], S[3]), s1 = make_float4(S[4], S[5], S[6], S[7]);
        const uint32_t r = fbits(S[8]);
        float t;
        const bool h = hit_sphere(s0, s1, o, d, time, tlo, best.t, r == origin_ref, t);
        best.t = h ? t : best.t;
        best.ref = h ? r : best.ref;
    }
    const RT_CONST_AS uint32_t *other = as_uniform(sc.flat_other);
    for (uint32_t k = 0; !BOXED && k < sc.n_flat_other; ++k) test_prim<true>(sc, other[k], o, d, time, tlo, origin_ref, rc, best);
}

template <int ACCEL, bool USE_LDS_NODES>
__device__ __forceinline__ void closest_hit(SceneRef sc, V3 o, V3 d, float time, float tlo, float thi, uint32_t origin_ref,
                                            const RayCtx &rc, uint32_t *s_stack, const rt_node *s_nodes, uint32_t *s_dyn,
                                            Best &best) {
    if (ACCEL == ACCEL_FLAT)
        scan_flat(sc, o, d, time, tlo, thi, origin_ref, rc, best);
    else if (ACCEL == ACCEL_BVH2)
        traverse<USE_LDS_NODES>(sc, o, d, time, tlo, thi, origin_ref, rc, s_stack + threadIdx.x, s_nodes, best);
    else
        if (sc.lds_nodes4) // uniform
            traverse4<true, ACCEL == ACCEL_BVH4Q>(sc, o, d, time, tlo, thi, origin_ref, rc, s_dyn, best);
        else
            traverse4<false, ACCEL == ACCEL_BVH4Q>(sc, o, d, time, tlo, thi, origin_ref, rc, s_dyn, best);
}

// LDS the three structures need, declared once per kernel
#define RT_ACCEL_LDS(ACCEL)                                                                  \
    __shared__ uint32_t s_stack[(ACCEL) == ACCEL_BVH2 ? STACK_DEPTH * BLOCK : 1];            \
    __shared__ rt_node s_nodes[(ACCEL) == ACCEL_BVH2 ? LDS_NODES : 1];                       \
    extern __shared__ uint32_t s_dyn[];

// ------------------------------------------------------------------ textures (texture.rs)
__device__ __forceinline__ float perlin_noise(const float4 *vec, const int *perm, V3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = int(fx), j = int(fy), k = int(fz);
    float uu = u * u * (3.f - 2.f * u), vv = v * v * (3.f - 2.f * v), ww = w * w * (3.f - 2.f * w);
    float accum = 0.f;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                float4 c = vec[perm[(i + di) & 255] ^ perm[256 + ((j + dj) & 255)] ^ perm[512 + ((k + dk) & 255)]];
                float wx = u - di, wy = v - dj, wz = w - dk;
                accum += (di ? uu : 1.f - uu) * (dj ? vv : 1.f - vv) * (dk ? ww : 1.f - ww) *
                         (c.x * wx + c.y * wy + c.z * wz);
            }
    return accum;
}

__device__ __forceinline__ V3 tex_value(SceneRef sc, uint32_t id, float u, float v, V3 p) {
    // Checkerboard children are arbitrary textures (texture.rs:39-43): follow the chain
    for (int depth = 0; depth < 16; ++depth) {
        rt_texture t = sc.texs[id];
        if (t.type == RT_TEX_SOLID) return {t.color[0], t.color[1], t.color[2]};
        if (t.type == RT_TEX_CHECKER) { // texture.rs:59-72
            int x = int(floorf(t.scale * p.x));
            int y = int(floorf(t.scale * p.y));
            int z = int(floorf(t.scale * p.z));
            bool even = ((x + y + z) % 2) == 0;
            id = even ? t.a : t.b;
            continue;
        }
        if (t.type == RT_TEX_IMAGE) { // texture.rs:125-139, texel index clamped to the image
            float uc = fminf(fmaxf(u, 0.f), 1.f);
            float vc = 1.0f - fminf(fmaxf(v, 0.f), 1.f);
            uint32_t i = uint32_t(uc * float(t.a));
            uint32_t j = uint32_t(vc * float(t.b));
            i = min(i, t.a - 1);
            j = min(j, t.b - 1);
            const uint8_t *px = sc.texels + t.offset + (size_t(j) * t.a + i) * 3;
            const float s = 1.0f / 255.0f;
            return {float(px[0]) * s, float(px[1]) * s, float(px[2]) * s};
        }
        // noise: marble of "The Next Week" ch.5 (no counterpart in the reference)
        const float4 *vec = sc.noise_vec + size_t(t.offset) * 256;
        const int *perm = sc.noise_perm + size_t(t.offset) * 768;
        float accum = 0.f, weight = 1.f;
        V3 q = p;
        for (int k = 0; k < 7; ++k) {
            accum += weight * perlin_noise(vec, perm, q);
            weight *= 0.5f;
            q = q * 2.0f;
        }
        float s = 0.5f * (1.0f + sinf(t.scale * p.z + 10.0f * fabsf(accum)));
        return {s, s, s};
    }
    return {0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------ lights (hittable.rs:377-395)
// Copies in LDS of what shading reads per vertex from a small scene -- a quad's normal and material
// id, the sphere records, the material records -- made once per block by k_paths_regroup: a
// vertex then costs two LDS reads instead of two dependent loads from memory (measured on the
// Book-3 Cornell box: the ray step's material lookup alone was 6.6 % of the frame).
constexpr uint32_t TABLE_QUADS = 48, TABLE_SPHERES = 8, TABLE_MATERIALS = 16;
struct FlatTables {
    const float4 *quad_nm;    // n.xyz, material id (bits)
    const float4 *spheres;    // as DScene::spheres
    const rt_material *mats;
};
// The light list, one record of 8 float4 per light in list order, made at upload from the lights'
// primitives so that sampling and the pdf read a light with one request (by reference it was two
// dependent ones: the ref, then the primitive):
//   L0 = Q | radius 0        or  centre | radius (> 0: a sphere)
//   L1 = n | area   L2 = avec   L3 = bvec   L4 = u   L5 = v   (parallelograms; the only quad shape
//   with pdf_value and random, hittable.rs:99-107, 377-395)
// Direction to a random point of light i, unnormalised (Parallelogram::random), or towards the
// visible cap of a sphere.  UNIFORM: i is the same in every lane, so the record comes through scalar loads.
template <bool UNIFORM>
__device__ __forceinline__ V3 light_random(SceneRef sc, uint32_t i, V3 p, Rng &rng) {
    const float4 *L = sc.light_recs + 8 * size_t(i);
    const float4 l0 = load4<UNIFORM>(L, 0), lu = load4<UNIFORM>(L, 4), lv = load4<UNIFORM>(L, 5); // one round trip
    if (!(l0.w > 0.0f)) {
        float a = rng.next();
        float b = rng.next();
        V3 lp = fma3(mk(lv), b, fma3(mk(lu), a, mk(l0)));
        return lp - p;
    }
    V3 dir = mk(l0) - p;
    float dist2 = len2(dir);
    ONB uvw(dir);
    float r1 = rng.next();
    float r2 = rng.next();
    float z = 1.0f + r2 * (fsqrt(1.0f - l0.w * l0.w * rcp(dist2)) - 1.0f);
    float sn = __builtin_amdgcn_sinf(r1), cs = __builtin_amdgcn_cosf(r1);
    float s = fsqrt(fmaxf(0.f, 1.0f - z * z));
    return uvw.transform(V3{cs * s, sn * s, z});
}

// Parallelogram::pdf_value (hittable.rs:377-389) / sphere cone pdf
template <bool UNIFORM>
__device__ __forceinline__ float light_pdf(SceneRef sc, uint32_t i, V3 p, V3 dir) {
    const float4 *L = sc.light_recs + 8 * size_t(i);
    const float4 q0 = load4<UNIFORM>(L, 0), q1 = load4<UNIFORM>(L, 1), q2 = load4<UNIFORM>(L, 2), q3 = load4<UNIFORM>(L, 3);
    if (!(q0.w > 0.0f)) {
        // the plane test of hit_quad without the shape cases
        const float denom = dot(mk(q1), dir);
        const V3 qo = mk(q0) - p;
        const float t = dot(mk(q1), qo) * rcp(denom);
        const V3 ph = fma3(dir, t, -qo);
        const float a = dot(ph, mk(q2)), b = dot(ph, mk(q3));
        const float outside = fmaxf(fmaxf(-a, -b), fmaxf(a, b) - 1.0f);
        if (!(fmaxf(outside, 0.001f - t) <= 0.0f) || !(fabsf(denom) >= 1e-8f)) return 0.0f;
        const float l2 = len2(dir);
        return t * t * l2 * rcp(fabsf(denom) * rsq(l2) * q1.w); // dist^2 / (cosine * area)
    }
    // Only hit or miss matters here, for an origin away from the light: the quadratic about the
    // point of closest approach, disc/a = r^2 - |oc - (h/a) d|^2, without the compensated terms
    // of hit_sphere.  A root lies in (0.001, inf) iff the far root does.
    const float4 s0 = q0;
    const V3 oc = mk(s0) - p; // pdf_value asks at time 0 (hittable.rs:378)
    const float a = len2(dir), h = dot(dir, oc), ia = rcp(a);
    const float disc = fmaf(s0.w, s0.w, -len2(fma3(dir, -(h * ia), oc)));
    if (!(disc >= 0.0f)) return 0.0f;
    if (!((h + fsqrt(a * disc)) * ia > 0.001f)) return 0.0f;
    const float cos_max = fsqrt(1.0f - s0.w * s0.w * rcp(len2(oc)));
    return rcp(2.0f * RT_PI * (1.0f - cos_max));
}

template <uint32_t LIGHTS = LIGHTS_ANY>
__device__ __forceinline__ V3 lights_random(SceneRef sc, V3 p, Rng &rng) {
    if (RT_LKIND(sc) == RT_LIGHTS_LIST) {
        uint32_t i = uint32_t(rng.next() * float(sc.n_lights));
        return light_random<false>(sc, min(i, sc.n_lights - 1), p, rng);
    }
    return light_random<true>(sc, 0u, p, rng);
}
template <uint32_t LIGHTS = LIGHTS_ANY>
__device__ __forceinline__ float lights_pdf(SceneRef sc, V3 p, V3 dir) {
    if (RT_LKIND(sc) != RT_LIGHTS_LIST) return light_pdf<true>(sc, 0u, p, dir);
    float w = rcp(float(sc.n_lights)), sum = 0.f;
    for (uint32_t i = 0; i < sc.n_lights; ++i) sum += w * light_pdf<true>(sc, i, p, dir);
    return sum;
}

// light_random in two parts, for a caller that samples a cone and a cosine lobe through ONE local frame:
// a parallelogram gives the direction itself (returns false); a sphere gives the frame's axis (unit),
// the azimuth and the local coordinates (s, z) of the cone sample -- the numbers light_random puts
// through its own basis, bit for bit.
template <bool UNIFORM>
__device__ __forceinline__ bool light_random_parts(SceneRef sc, uint32_t i, V3 p, Rng &rng, V3 &dir, V3 &axis, float &phi, float &s, float &z) {
    const float4 *L = sc.light_recs + 8 * size_t(i);
    const float4 l0 = load4<UNIFORM>(L, 0), lu = load4<UNIFORM>(L, 4), lv = load4<UNIFORM>(L, 5); // one round trip
    const float r1 = rng.next();
    const float r2 = rng.next();
    if (!(l0.w > 0.0f)) {
        dir = fma3(mk(lv), r2, fma3(mk(lu), r1, mk(l0))) - p;
        return false;
    }
    const V3 to = mk(l0) - p;
    axis = unit(to);
    z = 1.0f + r2 * (fsqrt(1.0f - l0.w * l0.w * rcp(len2(to))) - 1.0f);
    phi = r1;
    s = fsqrt(fmaxf(0.f, 1.0f - z * z));
    return true;
}

// ------------------------------------------------------------------ camera (camera.rs:409-430)
// rng: the (pixel, stratum) camera stream, begun by the caller
// PINHOLE: the caller knows that the camera has neither a defocus disk nor the disc filter (both are
// rejection loops with a Philox block inside)
template <bool PINHOLE = false>
__device__ __forceinline__ void camera_ray(CamRef cam, uint32_t pixel, uint32_t stratum, Rng &rng, V3 &o,
                                           V3 &d, float &time) {
    // row and column by a float reciprocal with one correction step: exact below 2^24, where
    // the integer division costs ~25 instructions apiece
    uint32_t i, j, si, sj;
    if (cam.npix < (1u << 24) && cam.sqrt_spp < 4096u) {
        j = uint32_t(float(pixel) * cam.inv_w);
        int32_t r = int32_t(pixel - j * cam.W);
        j += r >= int32_t(cam.W) ? 1u : 0u;
        j -= r < 0 ? 1u : 0u;
        i = pixel - j * cam.W;
        sj = uint32_t(float(stratum) * cam.inv_sqrt);
        int32_t rs = int32_t(stratum - sj * cam.sqrt_spp);
        sj += rs >= int32_t(cam.sqrt_spp) ? 1u : 0u;
        sj -= rs < 0 ? 1u : 0u;
        si = stratum - sj * cam.sqrt_spp;
    } else {
        i = pixel % cam.W, j = pixel / cam.W;
        si = stratum % cam.sqrt_spp, sj = stratum / cam.sqrt_spp;
    }
    float ox, oy;
    if (!PINHOLE && cam.aa_disc) { // AntialiasingType::Disc: r = 0.5 px about the pixel centre, not stratified
        for (;;) {
            ox = rng.pm1();
            oy = rng.pm1();
            if (ox * ox + oy * oy < 1.0f) break;
        }
        ox *= 0.5f, oy *= 0.5f;
    } else { // camera.rs:411-412
        ox = ((float(si) + rng.next()) * cam.inv_sqrt) - 0.5f;
        oy = ((float(sj) + rng.next()) * cam.inv_sqrt) - 0.5f;
    }
    // px_sample - camera_center, accumulated from small terms so the sub-pixel offset
    // survives fp32 (pixel_00 itself sits at scene-scale coordinates)
    V3 rel = fma3(ldv(cam.du), float(i), ldv(cam.p00_rel));
    rel = fma3(ldv(cam.dv), float(j), rel);
    rel = fma3(ldv(cam.du), ox, rel);
    rel = fma3(ldv(cam.dv), oy, rel);
    o = ldv(cam.center);
    d = rel;
    if (!PINHOLE && cam.defocus) { // camera.rs:484-488, two_d.rs:183-190
        float px, py;
        for (;;) {
            px = rng.pm1();
            py = rng.pm1();
            if (px * px + py * py < 1.0f) break;
        }
        V3 off = px * ldv(cam.disk_u) + py * ldv(cam.disk_v);
        o = ldv(cam.center) + off;
        d = rel - off;
    }
    time = cam.timeless ? 0.0f : rng.next(); // timeless: a render of a scene in which nothing moves
}
__device__ __forceinline__ void camera_ray(CamRef cam, uint32_t pixel, uint32_t stratum, uint32_t k0,
                                           uint32_t k1, V3 &o, V3 &d, float &time) {
    Rng rng;
    rng.begin(k0, k1, pixel, stratum, STREAM_CAMERA, 0);
    camera_ray(cam, pixel, stratum, rng, o, d, time);
}

// ------------------------------------------------------------------ kernels

// One block.  Exclusive prefix over the chunks' free slots -> sample ranges for generate.
__global__ void __launch_bounds__(1024) k_scan(Queue q, Ctrl *ctrl) {
    __shared__ uint32_t s_free[1024], s_alive[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (q.n_chunks + 1023u) / 1024u;
    const uint32_t lo = tid * per, hi = min(lo + per, q.n_chunks);
    uint32_t fsum = 0, asum = 0;
    for (uint32_t w = lo; w < hi; ++w) {
        uint32_t c = q.count[w];
        asum += c;
        fsum += q.cap - c;
    }
    s_free[tid] = fsum;
    s_alive[tid] = asum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) { // inclusive Hillis-Steele scan
        uint32_t f = 0, a = 0;
        if (tid >= off) {
            f = s_free[tid - off];
            a = s_alive[tid - off];
        }
        __syncthreads();
        s_free[tid] += f;
        s_alive[tid] += a;
        __syncthreads();
    }
    const unsigned long long next = ctrl->next_sample;
    const unsigned long long remaining = ctrl->total_samples - next;
    unsigned long long base = s_free[tid] - fsum; // exclusive prefix
    for (uint32_t w = lo; w < hi; ++w) {
        uint32_t fr = q.cap - q.count[w];
        unsigned long long left = remaining > base ? remaining - base : 0ull;
        uint32_t n = left < fr ? uint32_t(left) : fr;
        q.gen_base[w] = next + base;
        q.gen_n[w] = n;
        base += fr;
    }
    __syncthreads();
    if (tid == 1023) {
        unsigned long long total_free = s_free[1023];
        unsigned long long total_new = total_free < remaining ? total_free : remaining;
        uint32_t active = s_alive[1023] + uint32_t(total_new);
        ctrl->next_sample = next + total_new;
        ctrl->active = active;
        ctrl->segments += active;
        ctrl->camera_rays += total_new;
        ctrl->iterations += 1;
    }
}

__global__ void __launch_bounds__(BLOCK) k_generate(Queue q, const DCamera *camp, uint32_t k0, uint32_t k1) {
    CamRef cam = *as_uniform(camp);
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t chunk = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x / WAVE);
    if (chunk >= q.n_chunks) return;
    const uint32_t n = q.gen_n[chunk];
    if (n == 0) return;
    const uint32_t have = q.count[chunk];
    const unsigned long long gbase = q.gen_base[chunk];
    const size_t cbase = size_t(chunk) * q.cap;
    for (uint32_t k = lane; k < n; k += WAVE) {
        unsigned long long g = gbase + k;
        uint32_t pixel = uint32_t(g % cam.npix);
        uint32_t stratum = cam.stratum_begin + uint32_t(g / cam.npix);
        V3 o, d;
        float time;
        camera_ray(cam, pixel, stratum, k0, k1, o, d, time);
        size_t s = cbase + have + k;
        q.ra[s] = make_float4(o.x, o.y, o.z, time);
        q.rb[s] = make_float4(d.x, d.y, d.z, bitsf(RT_REF_NONE));
        q.sa[s] = make_float4(1.f, 1.f, 1.f, bitsf(pixel));
        q.sb[s] = make_uint2(stratum, 0u);
    }
    if (lane == 0) q.count[chunk] = have + n;
}

template <int ACCEL>
__global__ void __launch_bounds__(BLOCK) k_extend(Queue q, const DScene *scp, uint32_t k0, uint32_t k1, uint32_t has_media) {
    SceneRef sc = *as_uniform(scp);
    RT_ACCEL_LDS(ACCEL)
    if (accel_is_wide(ACCEL)) stage_nodes4<ACCEL == ACCEL_BVH4Q>(sc, s_dyn);
    if (ACCEL == ACCEL_BVH2) { // stage the top of the BVH
        const uint32_t nn = min(sc.n_nodes, uint32_t(LDS_NODES));
        const float4 *src = reinterpret_cast<const float4 *>(sc.nodes);
        float4 *dst = reinterpret_cast<float4 *>(s_nodes);
        for (uint32_t i = threadIdx.x; i < nn * 4; i += BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t chunk = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x / WAVE);
    if (chunk >= q.n_chunks) return;
    const uint32_t n = q.count[chunk];
    const size_t cbase = size_t(chunk) * q.cap;
    for (uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t k = base + lane;
        if (k >= n) continue;
        const size_t s = cbase + k;
        float4 ra = q.ra[s], rb = q.rb[s];
        V3 o = mk(ra), d = mk(rb);
        RayCtx rc{k0, k1, 0u, 0u, 0u};
        if (has_media) {
            rc.pixel = fbits(q.sa[s].w);
            uint2 sb = q.sb[s];
            rc.sample = sb.x;
            rc.bounce = sb.y;
        }
        Best best;
        closest_hit<ACCEL, true>(sc, o, d, ra.w, 0.001f, RT_INF, fbits(rb.w), rc, s_stack, s_nodes, s_dyn, best);
        q.hit[s] = make_float4(best.t, bitsf(best.ref), best.u, best.v);
    }
}

// The three planes follow one another in memory: one pointer and the plane size instead of three
// pointers (the persistent kernels sit at the limit of scalar registers).
struct Film {
    float *base;
    uint32_t npix;
};
__device__ __forceinline__ void film_add(const Film &f, uint32_t pixel, V3 c) {
    if (c.x != 0.f) atomicAdd(f.base + pixel, c.x);
    if (c.y != 0.f) atomicAdd(f.base + (pixel + f.npix), c.y);
    if (c.z != 0.f) atomicAdd(f.base + (pixel + size_t(2) * f.npix), c.z);
}

// camera.rs:445-453 for one path vertex: what the hit itself contributes.  A miss deposits the
// background, a light its emission; both end the path (returns false).  Otherwise fills `sf`
// for the scatter step.
struct Surface {
    V3 p, n;
    float u, v;
    uint32_t ref;
    bool front;
    rt_material m;
};
// quad_uv_missing: the hit came from the flat scan, which does not carry a quad's (alpha, beta)
// through its loop; they are recomputed here for the materials that look at them.
// DEPOSIT = false: the caller knows the hit is a surface that scatters (k_paths_regroup's scatter
// steps); only `sf` is wanted.  TABLES: quads, spheres and materials come from `tb`.
// PLAIN: nothing moves, there is no medium and every material's texture is a colour folded into its record.
template <bool DEPOSIT = true, bool TABLES = false, bool EXTRAS = true, bool PLAIN = false>
__device__ __forceinline__ bool shade_emit(SceneRef sc, CamRef cam, const Film &film, const float4 &ra,
                                           const float4 &rb, const float4 &sa, float4 h, bool quad_uv_missing,
                                           Surface &sf, const FlatTables &tb = FlatTables{}) {
    const V3 o = mk(ra), d = mk(rb), thr = mk(sa);
    const float time = PLAIN ? 0.0f : ra.w;
    const uint32_t pixel = fbits(sa.w);
    const uint32_t ref = fbits(h.y);
    // one deposit site for the two ways a path ends here (background, emission)
    V3 deposit = {0.f, 0.f, 0.f};
    bool surface = false;
    if (DEPOSIT && ref == RT_REF_NONE) { // camera.rs:445-448, 473-482
        V3 bg = ldv(cam.bg);
        if (cam.sky) {
            float a = 0.5f * (unit(d).y + 1.0f);
            bg = (1.0f - a) * V3{1.f, 1.f, 1.f} + a * V3{0.5f, 0.7f, 1.0f};
        }
        deposit = bg;
    } else {
        const V3 p = fma3(d, h.x, o);
        const uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
        V3 n;
        uint32_t mat_id;
        float u = h.z, v = h.w, u_shift = 0.f;
        bool front = true;
        if (kind == RT_KIND_QUAD) {
            if (TABLES) {
                const float4 nm = tb.quad_nm[idx];
                n = mk(nm);
                mat_id = fbits(nm.w);
            } else {
                const float4 *Q = sc.quads + 6 * idx;
                n = mk(Q[1]);
                mat_id = fbits(Q[3].w);
            }
        } else if (kind == RT_KIND_SPHERE) {
            float4 s0, s1;
            if (TABLES) {
                s0 = tb.spheres[3 * idx], s1 = tb.spheres[3 * idx + 1];
                u_shift = tb.spheres[3 * idx + 2].z;
            } else {
                const float4 *S = sc.spheres + 3 * idx;
                s0 = S[0], s1 = S[1];
                u_shift = S[2].z;
            }
            V3 center = PLAIN ? mk(s0) : fma3(mk(s1), time, mk(s0));
            n = (p - center) * rcp(s0.w);
            mat_id = fbits(s1.w);
        } else { // ConstantMedium: arbitrary normal, front face (hittable.rs:877-885)
            n = {1.f, 0.f, 0.f};
            mat_id = PLAIN ? 0u : sc.media[idx].mat;
        }
        if (kind != RT_KIND_MEDIUM) { // HitRecord::from_incoming_ray (hittable.rs:59-79)
            front = dot(d, n) < 0.0f;
            if (!front) n = -n;
        }
        rt_material m;
        if (TABLES) m = tb.mats[mat_id];
        else m = sc.mats[mat_id];
        if (!PLAIN && kind == RT_KIND_SPHERE && m.needs_uv) sphere_uv(front ? n : -n, u_shift, u, v);
        // (a face of a box leaf of the wide tree comes without its (alpha, beta) too: NaN)
        if (!PLAIN && kind == RT_KIND_QUAD && m.needs_uv) {
            if (quad_uv_missing) quad_uv(sc.quads + 6 * idx, p, u, v);
            else if (EXTRAS && u != u) quad_uv_ray(sc.quads + 6 * idx, o, d, h.x, u, v);
        }
        if (DEPOSIT && m.type == RT_MAT_LIGHT) { // material.rs:78-92: emits on the front face, never scatters
            // m.pad != 0: the texture is a SolidColor whose value the upload copied into m.albedo
            if (front) deposit = (PLAIN || m.pad) ? V3{m.albedo[0], m.albedo[1], m.albedo[2]} : tex_value(sc, m.tex, u, v, p);
        } else {
            sf.p = p, sf.n = n, sf.u = u, sf.v = v, sf.ref = ref, sf.front = front, sf.m = m;
            surface = true;
        }
    }
    if (DEPOSIT && !surface) film_add(film, pixel, thr * deposit);
    return surface;
}

// camera.rs:455-470: scatter at a surface, choose the next direction, weigh it.  rng is the
// vertex's shading stream, begun by the caller.  Returns true if the path continues.
// ONLY: the caller has sorted vertices by material and this call sees one kind -- RT_MAT_METAL,
// RT_MAT_DIELECTRIC, or RT_MAT_LAMBERTIAN for the two materials that scatter through a PDF; the
// other branches are compiled out.  SCATTER_ANY: every material, chosen per lane.
constexpr uint32_t SCATTER_ANY = 0xFFu;
// ONE_BASIS: the cosine sample and a sphere light's cone sample go through one local frame (one basis,
// one sine and cosine, one transform for the lanes of both); the same numbers as the separate ones.
// PLAIN: see shade_emit (no Isotropic material either: it is the phase function of a medium).
// LIGHTS: DScene::lights_kind if the caller knows it (the branches on it are compiled out), else LIGHTS_ANY.
template <int MODE, uint32_t ONLY = SCATTER_ANY, bool ONE_BASIS = false, bool PLAIN = false, uint32_t LIGHTS = LIGHTS_ANY>
__device__ __forceinline__ bool shade_scatter(SceneRef sc, CamRef cam, float4 &ra, float4 &rb, float4 &sa,
                                              uint2 &sb, const Surface &sf, Rng &rng, Ctrl *ctrl) {
    const V3 d = mk(rb), p = sf.p, n = sf.n;
    V3 thr = mk(sa);
    const rt_material &m = sf.m;
    const uint32_t bounce = sb.y;
    V3 dir, att;
    float weight = 1.0f;

    if (ONLY == SCATTER_ANY ? m.type == RT_MAT_METAL : ONLY == RT_MAT_METAL) { // material.rs:115-131 (specular: attenuation * L(reflected))
        V3 refl = d - n * (2.0f * dot(d, n));
        dir = unit(refl);
        if (m.param > 0.f) dir = dir + m.param * random_in_unit_sphere(rng);
        if (dot(dir, n) < 0.0f) return false;
        att = {m.albedo[0], m.albedo[1], m.albedo[2]};
    } else if (ONLY == SCATTER_ANY ? m.type == RT_MAT_DIELECTRIC : ONLY == RT_MAT_DIELECTRIC) { // material.rs:154-182, vec.rs:289-295
        float ri = sf.front ? rcp(m.param) : m.param;
        V3 ud = unit(d);
        float cos_theta = fminf(-dot(ud, n), 1.0f);
        float sin_theta = fsqrt(fmaxf(0.f, 1.0f - cos_theta * cos_theta));
        bool reflect_it = ri * sin_theta > 1.0f;
        if (!reflect_it) {
            float r0 = (1.0f - ri) * rcp(1.0f + ri);
            r0 = r0 * r0;
            float x = 1.0f - cos_theta;
            float x2 = x * x;
            reflect_it = (r0 + (1.0f - r0) * x2 * x2 * x) > rng.next();
        }
        if (reflect_it) {
            dir = ud - n * (2.0f * dot(ud, n));
        } else {
            V3 perp = ri * (ud + cos_theta * n);
            V3 par = n * (-fsqrt(fabsf(1.0f - len2(perp))));
            dir = perp + par;
        }
        att = {1.f, 1.f, 1.f};
    } else {
        // Lambertian (material.rs:45-63) or Isotropic (material.rs:198-212)
        const bool iso = !PLAIN && m.type == RT_MAT_ISOTROPIC;
        att = (PLAIN || m.pad) ? V3{m.albedo[0], m.albedo[1], m.albedo[2]} : tex_value(sc, m.tex, sf.u, sf.v, p);
        float spdf, pdf;
        if (MODE == RT_MODE_REF_EXACT) {
            // Shipped estimator: the material's own sample only feeds scattering_pdf
            // (camera.rs:464); the traced direction comes from the light (camera.rs:460-462).
            if (iso) {
                (void)random_in_unit_sphere(rng);
                spdf = 0.25f * RT_INV_PI;
            } else {
                (void)rng.next(); // r1: azimuth of the discarded cosine sample
                float r2 = rng.next();
                // n . unit(onb.transform(cosine sample)) = sqrt(1-r2)
                spdf = fmaxf(0.f, fsqrt(1.0f - r2) * RT_INV_PI);
            }
            dir = lights_random<LIGHTS>(sc, p, rng);
            pdf = lights_pdf<LIGHTS>(sc, p, dir);
        } else {
            // 0.5 light list + 0.5 material PDF (pdf.rs:62-90), intended branch order
            bool pick_light = false;
            if (RT_LKIND(sc) != RT_LIGHTS_NONE) pick_light = rng.next() < 0.5f;
            if (ONE_BASIS) {
                V3 axis = n;
                float phi = 0.f, sr = 0.f, z = 0.f;
                bool framed = true;
                if (pick_light) {
                    if (RT_LKIND(sc) == RT_LIGHTS_LIST) {
                        const uint32_t i = uint32_t(rng.next() * float(sc.n_lights));
                        framed = light_random_parts<false>(sc, min(i, sc.n_lights - 1), p, rng, dir, axis, phi, sr, z);
                    } else {
                        framed = light_random_parts<true>(sc, 0u, p, rng, dir, axis, phi, sr, z);
                    }
                } else if (iso) {
                    dir = random_in_unit_sphere(rng);
                    framed = false;
                } else {
                    phi = rng.next();
                    const float r2 = rng.next();
                    sr = fsqrt(r2);
                    z = fsqrt(1.0f - r2);
                }
                if (framed) {
                    const ONB uvw(axis, true);
                    dir = uvw.transform(V3{__builtin_amdgcn_cosf(phi) * sr, __builtin_amdgcn_sinf(phi) * sr, z});
                }
            } else {
            ONB uvw(n, true);
            if (pick_light) {
                dir = lights_random<LIGHTS>(sc, p, rng);
            } else if (iso) {
                dir = random_in_unit_sphere(rng);
            } else {
                float r1 = rng.next();
                float r2 = rng.next();
                float sr = fsqrt(r2);
                dir = uvw.transform(V3{__builtin_amdgcn_cosf(r1) * sr, __builtin_amdgcn_sinf(r1) * sr,
                                       fsqrt(1.0f - r2)});
            }
            }
            float cosn = dot(dir, n) * rsq(len2(dir)); // (n is the basis' w)
            float mpdf = iso ? 0.25f * RT_INV_PI : fmaxf(0.f, cosn * RT_INV_PI);
            spdf = mpdf; // Lambertian::scattering_pdf == CosinePDF::value; Isotropic likewise
            pdf = RT_LKIND(sc) != RT_LIGHTS_NONE ? 0.5f * lights_pdf<LIGHTS>(sc, p, dir) + 0.5f * mpdf : mpdf;
        }
        weight = spdf * rcp(pdf);
        if (!(pdf > 0.0f) || !(fabsf(weight) <= 3.0e38f)) { // would be inf/NaN radiance
            atomicAdd(&ctrl->dropped, 1ull);
            return false;
        }
        if (weight == 0.0f) return false; // the sample carries nothing further
    }
    if (bounce + 1 >= cam.max_depth) return false; // depth == 0 -> black (camera.rs:439-442)
    thr = thr * att * weight;
    ra = make_float4(p.x, p.y, p.z, ra.w);
    rb = make_float4(dir.x, dir.y, dir.z, bitsf(sf.ref));
    sa = make_float4(thr.x, thr.y, thr.z, sa.w);
    sb.y = bounce + 1;
    return true;
}

// camera.rs:451-470 for one path vertex.  Returns true if the path continues.
template <int MODE>
__device__ __forceinline__ bool shade_one(SceneRef sc, CamRef cam, const Film &film, uint32_t k0,
                                          uint32_t k1, float4 &ra, float4 &rb, float4 &sa, uint2 &sb, float4 h,
                                          bool quad_uv_missing, Ctrl *ctrl) {
    Surface sf;
    if (!shade_emit(sc, cam, film, ra, rb, sa, h, quad_uv_missing, sf)) return false;
    Rng rng;
    rng.begin(k0, k1, fbits(sa.w), sb.x, STREAM_SHADE, sb.y);
    return shade_scatter<MODE>(sc, cam, ra, rb, sa, sb, sf, rng, ctrl);
}

template <int MODE>
__global__ void __launch_bounds__(BLOCK) k_shade(Queue q, const DScene *scp, const DCamera *camp, Film film, uint32_t k0,
                                                  uint32_t k1, Ctrl *ctrl) {
    SceneRef sc = *as_uniform(scp);
    CamRef cam = *as_uniform(camp);
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t chunk = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x / WAVE);
    if (chunk >= q.n_chunks) return;
    const uint32_t n = q.count[chunk];
    if (n == 0) return;
    const size_t cbase = size_t(chunk) * q.cap;
    const bool flat_world = (sc.n_flat_pairs | sc.n_flat_quads | sc.n_flat_other | sc.n_flat_boxes | sc.n_flat_rects | sc.n_flat_spheres) != 0; // hits may come from the flat scan
    uint32_t out = 0; // survivors written so far; always <= base, so writes never pass reads
    for (uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t k = base + lane;
        bool alive = false;
        float4 ra, rb, sa;
        uint2 sb;
        if (k < n) {
            const size_t s = cbase + k;
            ra = q.ra[s];
            rb = q.rb[s];
            sa = q.sa[s];
            sb = q.sb[s];
            float4 h = q.hit[s];
            alive = shade_one<MODE>(sc, cam, film, k0, k1, ra, rb, sa, sb, h, flat_world, ctrl);
        }
        // wave-level stream compaction: ballot + mbcnt prefix
        const unsigned long long mask = __ballot(alive);
        const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
        if (alive) {
            const size_t s = cbase + out + prefix;
            q.ra[s] = ra;
            q.rb[s] = rb;
            q.sa[s] = sa;
            q.sb[s] = sb;
        }
        out += uint32_t(__popcll(mask));
    }
    if (lane == 0) q.count[chunk] = out;
}

// ------------------------------------------------------------------ persistent path kernel
// The same pixel x stratum loop with no ray queue: every lane carries one path in registers
// from its camera ray to its end, and a lane whose path has ended takes the next sample.  A
// wave owns a range of sample ids, taken up to `grab` at a time from one global counter, so the
// counter sees about one atomic per `grab` paths.  Dead lanes are refilled once at least
// `refill_min` of the 64 are idle (the camera ray costs a Philox block; refilling a few lanes
// at a time would run it at low lane occupancy every bounce).  Because every random number is
// keyed by (pixel, stratum, bounce), the image does not depend on which lane traces what.
// Exit: a wave leaves when the counter has passed total_samples and its lanes are idle; both
// are reached by every wave since a path ends after at most max_depth bounces.
// Smallest range a wave asks the counter for on a flat world.  All waves add to one address, which
// takes ~90 M atomics/s; there a wave uses up 2-5 samples per microsecond, so with 4096 waves
// asking, ranges below ~250 samples make the counter the bottleneck (a 0.7 ms frame took 0.25 ms
// longer with 64).  Behind a BVH a sample costs 10-100 times more and the smallest range stays
// one wave's worth, for the balance at the end of the frame.
constexpr uint32_t MIN_TAKE_FLAT = 256;
__device__ __forceinline__ unsigned long long wave_first(unsigned long long x) {
    uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(x));
    uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(x >> 32));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// Registers are held to 128 per lane (4 waves per SIMD): measured faster than 3 waves with no
// spills or 5, 6, 8 with more, on every workload, flat and BVH.
// EXTRAS (wide tree): the scene has box leaves or primitives tested after the walk.
// SIMPLE (wide tree, with EXTRAS): the camera is a pinhole with the square filter and `lights` is one
// primitive -- what is left to compile out for the Book-2 final scene, which has everything else (+2 %).
// PLAIN (wide tree): nothing in the scene moves, it has no medium and no Isotropic material, and every
// texture is a colour folded into its material's record -- the code for all of that is compiled out
// (the kernel sits at its register limit: +5.5 % on the Book-1 final scene, which is such a scene).
template <int MODE, int ACCEL, bool EXTRAS = true, bool PLAIN = false, bool SIMPLE = false>
__global__ void __launch_bounds__(BLOCK, 4) k_paths(const DScene *scp, const DCamera *camp, Film film, uint32_t k0, uint32_t k1, Ctrl *ctrl,
                                                  uint32_t has_media, uint32_t grab, uint32_t refill_min) {
    SceneRef sc = *as_uniform(scp);
    CamRef cam = *as_uniform(camp);
    RT_ACCEL_LDS(ACCEL)
    if (accel_is_wide(ACCEL)) stage_nodes4<ACCEL == ACCEL_BVH4Q>(sc, s_dyn);
    if (ACCEL == ACCEL_BVH2) {
        const uint32_t nn = min(sc.n_nodes, uint32_t(LDS_NODES));
        const float4 *src = reinterpret_cast<const float4 *>(sc.nodes);
        float4 *dst = reinterpret_cast<float4 *>(s_nodes);
        for (uint32_t i = threadIdx.x; i < nn * 4; i += BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const unsigned long long total = ctrl->total_samples;
    unsigned long long next = 0, end = 0; // sample ids this wave still owns (wave-uniform)
    unsigned long long seen = 0;          // counter value at this wave's last request
    const float inv_share = 1.0f / (4.0f * float(gridDim.x * (BLOCK / WAVE))); // a quarter of an even share
    uint32_t next_pixel = 0, next_stratum = 0; // (pixel, stratum) of sample id `next`
    unsigned long long n_seg = 0, n_cam = 0;
    bool exhausted = false;
    bool alive = false; // this lane holds a ray to trace
    float4 ra = {}, rb = {}, sa = {};
    uint2 sb = {};
    // Wide BVH only: a lane's traversal is resumable (trav4_round), so the wave does not wait for
    // its slowest ray before shading: once SHADE_QUORUM lanes have something to shade or a sample
    // to take, they do, and join the lanes still traversing afterwards.  Traversal lengths are
    // heavy-tailed; waiting for the last lane left 12 % of the lanes active on a large scene.
    uint32_t cur = RT_REF_NONE, sp = 0; // cur != RT_REF_NONE: this lane is inside a traversal
    Best best;
    best.t = RT_INF, best.ref = RT_REF_NONE, best.u = best.v = 0.f;
#ifdef RT_REGROUP_STATS
    // wide BVH, per wave: 0 node steps, 1 primitive steps, 2 shading passes (lanes with a hit to look at),
    // 3 lanes that scatter, 4 lanes that take a new sample, 5 calls for traversal rounds (lanes traversing)
    TravStats ts;
    unsigned long long pk_turns[RG_STATS] = {}, pk_lanes[RG_STATS] = {}, pk_ticks[RG_STATS] = {};
    unsigned long long pk_t0 = wall_clock64();
#define RT_PK_COUNT(i, m) (pk_turns[i] += 1, pk_lanes[i] += (m))
#define RT_PK_TICK(i) { const unsigned long long now_ = wall_clock64(); pk_ticks[i] += now_ - pk_t0; pk_t0 = now_; }
#define RT_PK_TS , &ts
#else
#define RT_PK_COUNT(i, m) ((void)0)
#define RT_PK_TICK(i) ((void)0)
#define RT_PK_TS
#endif
    for (;;) {
        bool scatter = false; // the path goes on from a surface
        bool busy = false;    // (wide BVH) this lane is inside a traversal and skips the shading steps
        Surface sf;
        if constexpr (accel_is_wide(ACCEL)) {
            busy = alive && cur != RT_REF_NONE;
            const uint32_t n_busy = uint32_t(__popcll(__ballot(busy)));
            // lanes with something to do in the shading steps: a finished traversal, or an empty
            // lane while samples are left
            const uint32_t n_work = uint32_t(__popcll(__ballot(!busy && (alive || !exhausted))));
            if (n_busy != 0 && n_work < SHADE_QUORUM) {
                if (busy) {
                    RayCtx rc{k0, k1, fbits(sa.w), sb.x, sb.y};
                    const TravMem mem = block_trav_mem(sc, s_dyn, ACCEL == ACCEL_BVH4Q ? 64u : 128u);
                    // rounds until the lanes that finish bring the shading steps their quorum
                    auto more = [&](uint32_t n_trav, uint32_t n_fin) { return n_trav != 0 && n_work + n_fin < SHADE_QUORUM; };
                    if (sc.lds_nodes4) // uniform
                        trav4_rounds<true, ACCEL == ACCEL_BVH4Q, false, BLOCK, EXTRAS, PLAIN>(sc, mk(ra), mk(rb), ra.w, 0.001f, fbits(rb.w), rc, mem, cur, sp, best, more RT_PK_TS);
                    else
                        trav4_rounds<false, ACCEL == ACCEL_BVH4Q, false, BLOCK, EXTRAS, PLAIN>(sc, mk(ra), mk(rb), ra.w, 0.001f, fbits(rb.w), rc, mem, cur, sp, best, more RT_PK_TS);
                }
                RT_PK_COUNT(5, n_busy);
                RT_PK_TICK(5);
                continue;
            }
            if (__ballot(alive) == 0 && exhausted) break;
            RT_PK_COUNT(2, uint32_t(__popcll(__ballot(alive && !busy))));
            if (alive && !busy) {
                if (EXTRAS && sc.n_post) { // uniform
                    RayCtx rc{k0, k1, fbits(sa.w), sb.x, sb.y};
                    test_post(sc, mk(ra), mk(rb), ra.w, 0.001f, fbits(rb.w), rc, best);
                }
                // (a quad's (alpha, beta) are not carried through the walk: shade_emit makes them from the hit
                // point for the materials that look at them, as it does behind the flat scan -- two registers
                // per lane in a kernel at its limit)
                scatter = shade_emit<true, false, EXTRAS, PLAIN>(sc, cam, film, ra, rb, sa, make_float4(best.t, bitsf(best.ref), 0.f, 0.f),
                                                                 true, sf);
            }
            RT_PK_COUNT(3, uint32_t(__popcll(__ballot(scatter))));
            RT_PK_TICK(2); // post tests and what the hit itself contributes
        } else {
            // 1. closest hit, then what the hit itself contributes (background, emission)
            const unsigned long long live = __ballot(alive);
            if (live == 0 && exhausted) break;
            n_seg += uint32_t(__popcll(live));
            if (alive) {
                RayCtx rc{k0, k1, fbits(sa.w), sb.x, sb.y};
                closest_hit<ACCEL, true>(sc, mk(ra), mk(rb), ra.w, 0.001f, RT_INF, fbits(rb.w), rc, s_stack, s_nodes, s_dyn, best);
                scatter = shade_emit<true, false, false>(sc, cam, film, ra, rb, sa, make_float4(best.t, bitsf(best.ref), best.u, best.v),
                                                         ACCEL == ACCEL_FLAT, sf);
            }
        }
        // 2. idle lanes take new samples
        bool fresh = false;
        const unsigned long long idle = __ballot(!scatter && !busy);
        const uint32_t n_idle = uint32_t(__popcll(idle));
        if (!exhausted && (n_idle >= refill_min || n_idle == WAVE || (accel_is_wide(ACCEL) && n_idle != 0))) {
            if (next == end) {
                // guided self-scheduling: ranges shrink with the work that is left (estimated
                // from the counter value this wave saw last), so the last waves finish together
                const float left = float(total > seen ? total - seen : 0ull);
                uint32_t take = uint32_t(fminf(left * inv_share, float(grab)));
                const uint32_t min_take = ACCEL == ACCEL_FLAT ? MIN_TAKE_FLAT : uint32_t(WAVE);
                take = take < min_take ? min_take : (take & ~uint32_t(WAVE - 1));
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(&ctrl->next_sample, static_cast<unsigned long long>(take));
                b = wave_first(b);
                seen = b;
                next = b < total ? b : total;
                end = b + take < total ? b + take : total;
                exhausted = next == end;
                next_pixel = uint32_t(next % cam.npix); // the only 64-bit divisions: once per grab
                next_stratum = cam.stratum_begin + uint32_t(next / cam.npix);
            }
            const uint32_t prefix =
                __builtin_amdgcn_mbcnt_hi(uint32_t(idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle), 0u));
            const unsigned long long g = next + prefix;
            if (!scatter && !busy && g < end) {
                uint32_t px = next_pixel + prefix, st = next_stratum;
                while (px >= cam.npix) px -= cam.npix, ++st;
                sa.w = bitsf(px);
                sb = make_uint2(st, 0u);
                fresh = true;
            }
            const uint32_t took = uint32_t(end - next < n_idle ? end - next : n_idle);
            next += took;
            n_cam += took;
            next_pixel += took;
            while (next_pixel >= cam.npix) next_pixel -= cam.npix, ++next_stratum;
        }
        // 3. one Philox block for every lane: the camera stream of a new sample or the
        //    shading stream of this vertex
        Rng rng;
        rng.begin(k0, k1, fbits(sa.w), sb.x, fresh ? STREAM_CAMERA : STREAM_SHADE, sb.y);
        RT_PK_COUNT(4, uint32_t(__popcll(__ballot(fresh))));
        RT_PK_TICK(6); // new samples taken, one Philox block
        if (fresh) {
            V3 o, d;
            float time;
            camera_ray<SIMPLE>(cam, fbits(sa.w), sb.x, rng, o, d, time);
            ra = make_float4(o.x, o.y, o.z, time);
            rb = make_float4(d.x, d.y, d.z, bitsf(RT_REF_NONE));
            sa = make_float4(1.f, 1.f, 1.f, sa.w);
            alive = true;
        }
#ifdef RT_REGROUP_STATS // (the product build keeps one if / else chain: split like this it compiles 8 % slower)
        RT_PK_TICK(4); // camera rays
        if (!fresh && scatter) {
#else
        else if (scatter) {
#endif
            alive = shade_scatter<MODE, SCATTER_ANY, false, PLAIN, SIMPLE ? uint32_t(RT_LIGHTS_SINGLE) : LIGHTS_ANY>(sc, cam, ra, rb, sa, sb, sf, rng, ctrl);
        } else if (!fresh && !busy) {
            alive = false;
        }
        RT_PK_TICK(3); // scatter
        if constexpr (accel_is_wide(ACCEL)) { // lanes with a new ray start its traversal
            const bool start = alive && !busy;
            n_seg += uint32_t(__popcll(__ballot(start)));
            if (start) trav4_begin(RT_INF, cur, sp, best);
        }
    }
    if (lane == 0) {
        atomicAdd(&ctrl->segments, n_seg);
        atomicAdd(&ctrl->camera_rays, n_cam);
#ifdef RT_REGROUP_STATS
        pk_turns[0] = ts.node_steps, pk_lanes[0] = ts.node_lanes, pk_turns[1] = ts.prim_steps, pk_lanes[1] = ts.prim_lanes;
        for (uint32_t i = 0; i < RG_STATS; ++i)
            atomicAdd(&g_rg_stats[i], pk_turns[i]), atomicAdd(&g_rg_stats[RG_STATS + i], pk_lanes[i]), atomicAdd(&g_rg_stats[2 * RG_STATS + i], pk_ticks[i]);
#endif
    }
#undef RT_PK_COUNT
#undef RT_PK_TICK
#undef RT_PK_TS
}

// ------------------------------------------------------------------ persistent path kernel, regrouping
// Flat worlds.  The same loop as k_paths, but a path is not tied to a lane: every wave keeps a pool
// of REGROUP_SLOTS paths in LDS and, per pool, one list of slot numbers for every kind of work a
// path can be waiting for -- a closest hit to find, a scatter at a diffuse / metal / glass vertex,
// its end (what a light or the background contributes, then a new camera ray into the same slot).
// Each turn the wave takes up to 64 slots from the fullest list and runs that one step for them,
// so the step's code runs with most lanes active instead of the 10-20 of 64 that want it in a
// wave of lane-bound paths.  The pool belongs to one wave: lists are pushed with ballot + mbcnt
// prefixes, there are no barriers and no atomics.  Every random number is keyed by
// (pixel, stratum, bounce), so the image equals k_paths'.
// Exit: a wave leaves when every list is empty -- once the sample counter has run out, a slot whose
// path has ended gets no new sample and leaves the lists; each step moves its slots on and a path
// ends within max_depth bounces.
#ifndef RT_REGROUP_SLOTS
#define RT_REGROUP_SLOTS 172
#endif
#ifndef RT_REGROUP_SLOTS_TIMELESS
#define RT_REGROUP_SLOTS_TIMELESS 184
#endif
// slots of a wave's pool: in a scene in which nothing moves a slot does without the ray's time
constexpr int REGROUP_SLOTS = RT_REGROUP_SLOTS, REGROUP_SLOTS_TIMELESS = RT_REGROUP_SLOTS_TIMELESS;
enum : uint32_t { RG_FREE = 0, RG_RAY = 1, RG_DIFFUSE = 2, RG_METAL = 3, RG_GLASS = 4, RG_END = 5, RG_LISTS = 6 };
constexpr uint32_t RG_REF_NOTHING = 0xFFFFFFFEu; // in place of a hit's ref: the path ended with nothing to deposit, or the slot never held one
// A slot is 52 bytes + 5 list bytes (48 + 5 when nothing moves): the pool size sets how full the turns
// run (measured on the Book-3 Cornell box: 96 slots 34.8, 128 42.0, 144 43.8, 176 45.9 G segments/s),
// and the 16 waves of a compute unit share 160 KB: 4 x 172 (184) slots are 39 232 (39 040) bytes a
// block, the scene tables (FlatTables) take 1 728 more.
//   ra: origin and time; once the closest hit is found, the hit point and time
//   rb: direction and the ref of the primitive the ray leaves; once the hit is found, the hit's ref
//   sa: throughput and pixel;  sb: stratum | bounce << 24 (TIMELESS: in ra.w, where the time would be)
template <bool TIMELESS> struct alignas(16) RegroupPool {
    static constexpr int SLOTS = TIMELESS ? REGROUP_SLOTS_TIMELESS : REGROUP_SLOTS;
    float4 ra[SLOTS], rb[SLOTS], sa[SLOTS];
    uint32_t sb[TIMELESS ? 1 : SLOTS];
    uint8_t list[RG_LISTS - 1][SLOTS]; // list `which` at [which - 1]: a free slot waits in the end list (k_paths_regroup)
};
constexpr uint32_t REGROUP_MAX_STRATA = 1u << 24, REGROUP_MAX_DEPTH = 255u; // what sb holds
static_assert(REGROUP_SLOTS <= 256 && REGROUP_SLOTS >= WAVE && REGROUP_SLOTS_TIMELESS <= 256 && REGROUP_SLOTS_TIMELESS >= WAVE, "slot numbers are bytes");
#ifndef RT_REGROUP_WAVES
#define RT_REGROUP_WAVES 4
#endif
#ifndef RT_ONE_BASIS
#define RT_ONE_BASIS 1
#endif
static_assert(RG_STATS == RG_LISTS + 2, "the lists' steps, then harvest and refill of k_paths_pool");
#ifdef RT_REGROUP_STATS
#define RT_RG_COUNT(stage, m) (rg_turns[stage] += 1, rg_lanes[stage] += (m))
#else
#define RT_RG_COUNT(stage, m) ((void)0)
#endif

// TABLES: the scene is small enough for FlatTables (the launch checks the counts).
// TIMELESS: nothing in the scene moves, every ray's time is 0 (DCamera::timeless) and a slot does without it.
// BOXED: see scan_flat; moreover nothing moves (TIMELESS), there is no medium and no Isotropic material, and
// every texture is a colour folded into its material's record (PLAIN of shade_emit / shade_scatter).
// LIGHTS: see shade_scatter.
template <int MODE, bool TABLES, bool TIMELESS, bool BOXED = false, uint32_t LIGHTS = LIGHTS_ANY>
__global__ void __launch_bounds__(BLOCK, RT_REGROUP_WAVES) k_paths_regroup(const DScene *scp, const DCamera *camp, Film film, uint32_t k0,
                                                          uint32_t k1, Ctrl *ctrl, uint32_t has_media, uint32_t grab,
                                                          uint32_t n_quads, uint32_t n_spheres, uint32_t n_mats, uint32_t first_take) {
    SceneRef sc = *as_uniform(scp);
    CamRef cam = *as_uniform(camp);
    static_assert(!BOXED || (TIMELESS && TABLES), "the boxed variant is compiled for still scenes with tables");
    typedef RegroupPool<TIMELESS> Pool;
    constexpr uint32_t SLOTS = Pool::SLOTS;
    __shared__ Pool s_pool[BLOCK / WAVE];
    __shared__ float4 s_quad_nm[TABLES ? TABLE_QUADS : 1], s_spheres[TABLES ? 3 * TABLE_SPHERES : 1];
    __shared__ rt_material s_mats[TABLES ? TABLE_MATERIALS : 1];
    __shared__ uint8_t s_next[TABLES ? TABLE_QUADS + TABLE_SPHERES + 8 : 1];
    if (TABLES) { // the only barriers of the kernel: before any wave takes its first turn
        for (uint32_t i = threadIdx.x; i < n_quads; i += BLOCK) {
            const float4 q1 = sc.quads[6 * i + 1];
            s_quad_nm[i] = make_float4(q1.x, q1.y, q1.z, sc.quads[6 * i + 3].w);
        }
        for (uint32_t i = threadIdx.x; i < 3 * n_spheres; i += BLOCK) s_spheres[i] = sc.spheres[i];
        for (uint32_t i = threadIdx.x; i < n_mats; i += BLOCK) s_mats[i] = sc.mats[i];
        __syncthreads();
        // the list a path joins after a hit on quad i / sphere i, by its material's kind; the last entry is a miss
        auto list_of = [&](uint32_t mat_id) {
            const uint32_t type = s_mats[mat_id].type;
            return uint8_t(type == RT_MAT_LIGHT ? RG_END : (type == RT_MAT_METAL ? RG_METAL : (type == RT_MAT_DIELECTRIC ? RG_GLASS : RG_DIFFUSE)));
        };
        for (uint32_t i = threadIdx.x; i < n_quads; i += BLOCK) s_next[i] = list_of(fbits(s_quad_nm[i].w));
        for (uint32_t i = threadIdx.x; i < n_spheres; i += BLOCK) s_next[TABLE_QUADS + i] = list_of(fbits(s_spheres[3 * i + 1].w));
        if (threadIdx.x == 0) s_next[TABLE_QUADS + TABLE_SPHERES] = uint8_t(RG_END);
        __syncthreads();
    }
    const FlatTables tb{s_quad_nm, s_spheres, s_mats};
    Pool &pool = s_pool[threadIdx.x / WAVE];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    // A slot that holds no path waits in the end list with nothing to deposit: the end step gives it
    // the next sample, as it does to a slot whose path has just ended.  (A list of free slots beside it
    // was one more byte per slot and one more count to compare.)
    for (uint32_t i = lane; i < SLOTS; i += WAVE) {
        pool.list[RG_END - 1][i] = uint8_t(i);
        pool.rb[i].w = bitsf(RG_REF_NOTHING);
    }
    // list lengths, wave-uniform
    uint32_t n_ray = 0, n_diffuse = 0, n_metal = 0, n_glass = 0, n_end = SLOTS;
    // stratum | bounce << 24 of a slot: a word of its own, or (TIMELESS) the w of ra in place of the time
    auto get_sb = [&](uint32_t slot, const float4 &ra) { return TIMELESS ? fbits(ra.w) : pool.sb[slot]; };
    auto timed = [&](float4 ra) { // the slot's ra as the shading code reads it: w is the ray's time
        if (TIMELESS) ra.w = 0.0f;
        return ra;
    };
    // The wave owns a range of sample ids: `remaining` of them are left, the next one is (next_pixel,
    // next_stratum).  `left` is how many the whole frame had left when the wave last asked the counter.
    // (Counts instead of 64-bit ids: the kernel sits at the limit of scalar registers.)
    // A wave's first range comes with the launch: wave w owns ids [w, w + 1) * first_take, and what
    // the counter hands out starts behind all of them.  (Sixteen thousand waves asking one address
    // for their first range at once took 0.19 ms to be served: nothing in a 100 ms frame, close to
    // half of a 0.4 ms one.)  first_take = 0: every range from the counter.
    const unsigned long long total = ctrl->total_samples;
    const unsigned long long first_id = (unsigned long long)first_take * (blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE);
    uint32_t remaining = first_take;
    float left = float(total - (unsigned long long)first_take * (gridDim.x * (BLOCK / WAVE)));
    uint32_t next_pixel = uint32_t(first_id % cam.npix), next_stratum = cam.stratum_begin + uint32_t(first_id / cam.npix);
    // segments and camera rays since the wave last added them to the frame's totals: 32-bit counts,
    // handed over at a refill before they can overflow (a range is at most `grab` paths of at most 255 segments)
    uint32_t n_seg = 0, n_cam = 0;
    auto hand_over_counts = [&]() {
        if (lane == 0) {
            atomicAdd(&ctrl->segments, static_cast<unsigned long long>(n_seg));
            atomicAdd(&ctrl->camera_rays, static_cast<unsigned long long>(n_cam));
        }
        n_seg = n_cam = 0;
    };
    bool exhausted = false;
#ifdef RT_REGROUP_STATS
    unsigned long long rg_turns[RG_STATS] = {}, rg_lanes[RG_STATS] = {}, rg_ticks[RG_STATS] = {};
#endif
    // appends the slots of the lanes where `yes` holds to list `which`
    auto push = [&](uint32_t which, uint32_t &n, bool yes, uint32_t slot) {
        const unsigned long long mask = __ballot(yes);
        if (mask == 0) return; // uniform
        const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
        if (yes) pool.list[which - 1][n + prefix] = uint8_t(slot);
        n += uint32_t(__popcll(mask));
    };
    // up to `want` new samples from the wave's range (asking the counter for a new range when it
    // is used up): lanes below the returned count make their camera ray (camera.rs:409-430) in
    // `slot`, which joins the rays to trace
    auto new_paths = [&](uint32_t want, uint32_t slot) -> uint32_t {
        if (remaining == 0 && !exhausted) {
            if ((n_seg | n_cam) >> 30) hand_over_counts();
            const float inv_share = 1.0f / (4.0f * float(gridDim.x * (BLOCK / WAVE)));
            uint32_t take = uint32_t(fminf(left * inv_share, float(grab)));
            take = take < MIN_TAKE_FLAT ? MIN_TAKE_FLAT : (take & ~uint32_t(WAVE - 1));
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(&ctrl->next_sample, static_cast<unsigned long long>(take));
            b = wave_first(b) + (unsigned long long)first_take * (gridDim.x * (BLOCK / WAVE)); // behind the ranges the launch gave out
            left = float(total > b ? total - b : 0ull);
            const unsigned long long first = b < total ? b : total, last = b + take < total ? b + take : total;
            remaining = uint32_t(last - first);
            exhausted = remaining == 0;
            next_pixel = uint32_t(first % cam.npix);
            next_stratum = cam.stratum_begin + uint32_t(first / cam.npix);
        }
        const uint32_t m = remaining < want ? remaining : want;
        if (m == 0) return 0;
        if (lane < m) {
            uint32_t px = next_pixel + lane, st = next_stratum;
            while (px >= cam.npix) px -= cam.npix, ++st;
            Rng rng;
            rng.begin(k0, k1, px, st, STREAM_CAMERA, 0);
            V3 o, d;
            float time;
            camera_ray(cam, px, st, rng, o, d, time);
            pool.ra[slot] = make_float4(o.x, o.y, o.z, TIMELESS ? bitsf(st) : time);
            pool.rb[slot] = make_float4(d.x, d.y, d.z, bitsf(RT_REF_NONE));
            pool.sa[slot] = make_float4(1.f, 1.f, 1.f, bitsf(px));
            if (!TIMELESS) pool.sb[slot] = st;
            pool.list[RG_RAY - 1][n_ray + lane] = uint8_t(slot);
        }
        n_ray += m;
        remaining -= m;
        n_cam += m;
        next_pixel += m;
        while (next_pixel >= cam.npix) next_pixel -= cam.npix, ++next_stratum;
        return m;
    };
    for (;;) {
        // lanes of one wave run in step, so LDS written in one turn is read back in the next
        // without a barrier; the fence keeps the compiler from moving accesses across turns
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // the fullest list, 64 counting as full; among equally full lists the first of ray, diffuse, glass,
        // metal, end.  Most turns find a full list: those are spared the comparison of all six counts
        // (sixty-five scalar instructions a turn, two fifths of the kernel's scalar work).
        uint32_t stage = RG_RAY, most = uint32_t(WAVE);
#ifndef RT_NO_FAST_LIST_CHOICE
        if (n_ray >= uint32_t(WAVE)) stage = RG_RAY;
        else if (n_diffuse >= uint32_t(WAVE)) stage = RG_DIFFUSE;
        else if (n_glass >= uint32_t(WAVE)) stage = RG_GLASS;
        else if (n_metal >= uint32_t(WAVE)) stage = RG_METAL;
        else if (n_end >= uint32_t(WAVE)) stage = RG_END;
        else
#endif
        {
            most = min(n_ray, uint32_t(WAVE));
            auto offer = [&](uint32_t which, uint32_t n) {
                const uint32_t c = min(n, uint32_t(WAVE));
                if (c > most) most = c, stage = which;
            };
            offer(RG_DIFFUSE, n_diffuse);
            offer(RG_GLASS, n_glass);
            offer(RG_METAL, n_metal);
            offer(RG_END, n_end);
            if (most == 0) break;
        }
#ifdef RT_REGROUP_STATS
        const unsigned long long rg_t0 = wall_clock64();
        const uint32_t rg_stage = stage;
#endif

        if (stage == RG_END) {
            // what the path's last hit contributes -- background or emission (camera.rs:445-453)
            // -- and the next sample into the same slot
            const uint32_t m = most;
            RT_RG_COUNT(RG_END, m);
            n_end -= m;
            const bool active = lane < m;
            uint32_t slot = 0;
            if (active) {
                slot = pool.list[RG_END - 1][n_end + lane];
                const float4 rb = pool.rb[slot];
                if (fbits(rb.w) != RG_REF_NOTHING) {
                    Surface sf; // ra holds the hit point: t = 0 from there
                    (void)shade_emit<true, TABLES, true, BOXED>(sc, cam, film, timed(pool.ra[slot]), rb, pool.sa[slot], make_float4(0.f, rb.w, 0.f, 0.f), true, sf, tb);
                }
            }
            const uint32_t taken = new_paths(m, slot);
            // A slot that got no sample waits on with nothing to deposit until the wave's next range has
            // come; once the frame's samples have run out it leaves the lists for good.
            if (taken < m && !exhausted) {
                if (active && lane >= taken) pool.rb[slot].w = bitsf(RG_REF_NOTHING);
                push(RG_END, n_end, active && lane >= taken, slot);
            }
        } else if (stage == RG_RAY) {
            // closest hit (camera.rs:444)
            const uint32_t m = most;
            RT_RG_COUNT(RG_RAY, m);
            n_ray -= m;
            n_seg += m;
            const bool active = lane < m;
            uint32_t slot = 0, kind = RG_LISTS;
            if (active) {
                slot = pool.list[RG_RAY - 1][n_ray + lane];
                const float4 ra = pool.ra[slot], rb = pool.rb[slot];
                RayCtx rc{k0, k1, 0u, 0u, 0u};
                if (!BOXED && has_media) { // a medium keys its free-flight draw by the sample and the bounce
                    const uint32_t sb = get_sb(slot, ra);
                    rc.pixel = fbits(pool.sa[slot].w), rc.sample = sb & 0xFFFFFFu, rc.bounce = sb >> 24;
                }
                Best best;
                scan_flat<BOXED>(sc, mk(ra), mk(rb), TIMELESS ? 0.0f : ra.w, 0.001f, RT_INF, fbits(rb.w), rc, best);
                // the material's kind decides where the path waits next; a miss ends it like a light
                uint32_t type = RT_MAT_LIGHT;
                bool looked_up = false;
                if (TABLES && (BOXED || !has_media)) { // one table read by (kind, index); a miss reads the table's last entry
                    const uint32_t pk = best.ref >> RT_REF_KIND_SHIFT, idx = best.ref & RT_REF_INDEX_MASK;
                    const uint32_t at = best.ref == RT_REF_NONE ? TABLE_QUADS + TABLE_SPHERES : (pk == RT_KIND_SPHERE ? TABLE_QUADS + idx : idx);
                    kind = s_next[at];
                    looked_up = true;
                } else if (best.ref != RT_REF_NONE) {
                    const uint32_t pk = best.ref >> RT_REF_KIND_SHIFT, idx = best.ref & RT_REF_INDEX_MASK;
                    uint32_t mat_id;
                    if (TABLES) {
                        mat_id = pk == RT_KIND_QUAD ? fbits(s_quad_nm[idx].w) : pk == RT_KIND_SPHERE ? fbits(s_spheres[3 * idx + 1].w) : sc.media[idx].mat;
                        type = s_mats[mat_id].type;
                    } else {
                        mat_id = pk == RT_KIND_QUAD ? fbits(sc.quads[6 * idx + 3].w) : pk == RT_KIND_SPHERE ? fbits(sc.spheres[3 * idx + 1].w) : sc.media[idx].mat;
                        type = sc.mats[mat_id].type;
                    }
                }
                const V3 p = fma3(mk(rb), best.t, mk(ra)); // as shade_emit forms it
                pool.ra[slot] = make_float4(p.x, p.y, p.z, ra.w);
                pool.rb[slot].w = bitsf(best.ref);
                if (!looked_up)
                    kind = type == RT_MAT_LIGHT ? RG_END
                                                : (type == RT_MAT_METAL ? RG_METAL : (type == RT_MAT_DIELECTRIC ? RG_GLASS : RG_DIFFUSE));
            }
            push(RG_DIFFUSE, n_diffuse, kind == RG_DIFFUSE, slot);
            push(RG_END, n_end, kind == RG_END, slot);
            push(RG_GLASS, n_glass, kind == RG_GLASS, slot);
            push(RG_METAL, n_metal, kind == RG_METAL, slot);
        } else {
            // scatter at the vertices of one kind (camera.rs:455-470)
            const uint32_t m = most;
            auto scatter_turn = [&](auto only, uint32_t &n_list) {
                constexpr uint32_t ONLY = decltype(only)::value;
                RT_RG_COUNT(stage, m);
                n_list -= m;
                const bool active = lane < m;
                uint32_t slot = 0;
                bool alive = false, again = false; // again: the next hit is known and is a vertex of this kind
                if (active) {
                    slot = pool.list[stage - 1][n_list + lane];
                    float4 ra = pool.ra[slot], rb = pool.rb[slot], sa = pool.sa[slot];
                    const uint32_t packed = get_sb(slot, ra);
                    ra = timed(ra);
                    uint2 sb = make_uint2(packed & 0xFFFFFFu, packed >> 24);
                    Surface sf; // ra holds the hit point: t = 0 from there
                    (void)shade_emit<false, TABLES, true, BOXED>(sc, cam, film, ra, rb, sa, make_float4(0.f, rb.w, 0.f, 0.f), true, sf, tb);
                    Rng rng;
                    rng.begin(k0, k1, fbits(sa.w), sb.x, STREAM_SHADE, sb.y);
                    alive = shade_scatter<MODE, ONLY, RT_ONE_BASIS, BOXED, LIGHTS>(sc, cam, ra, rb, sa, sb, sf, rng, ctrl);
                    if (ONLY == RT_MAT_DIELECTRIC && alive && (sf.ref >> RT_REF_KIND_SHIFT) == RT_KIND_SPHERE) {
                        // A ray going into a sphere that nothing else reaches into (marked at
                        // upload) meets that sphere next: the closest-hit query is answered here,
                        // with the root the scan's own sphere test takes for a ray that starts on
                        // the sphere, and the path stays with the glass vertices.
                        const float4 *S = (TABLES ? s_spheres : sc.spheres) + 3 * (sf.ref & RT_REF_INDEX_MASK);
                        const V3 dir = mk(rb);
                        if (fbits(S[2].w) != 0u && (dot(dir, sf.n) < 0.0f) == sf.front) {
                            const SphereEq q = sphere_eq(S[0], S[1], mk(ra), dir, ra.w);
                            const float root = 2.0f * q.h * q.inv_a;
                            if (0.001f < root && root < RT_INF && !(q.disc < 0.0f)) {
                                const V3 hp = fma3(dir, root, mk(ra));
                                ra = make_float4(hp.x, hp.y, hp.z, ra.w);
                                again = true;
                            }
                        }
                    }
                    if (alive) {
                        if (TIMELESS) ra.w = bitsf(sb.x | (sb.y << 24));
                        else pool.sb[slot] = sb.x | (sb.y << 24);
                        pool.ra[slot] = ra;
                        pool.rb[slot] = rb;
                        pool.sa[slot] = sa;
                    } else {
                        pool.rb[slot].w = bitsf(RG_REF_NOTHING);
                    }
                }
                if (ONLY == RT_MAT_DIELECTRIC) {
                    n_seg += uint32_t(__popcll(__ballot(again)));
                    push(stage, n_list, again, slot);
                }
                push(RG_RAY, n_ray, alive && !again, slot);
                push(RG_END, n_end, active && !alive, slot);
            };
            if (stage == RG_DIFFUSE) scatter_turn(std::integral_constant<uint32_t, RT_MAT_LAMBERTIAN>{}, n_diffuse);
            else if (stage == RG_GLASS) scatter_turn(std::integral_constant<uint32_t, RT_MAT_DIELECTRIC>{}, n_glass);
            else scatter_turn(std::integral_constant<uint32_t, RT_MAT_METAL>{}, n_metal);
        }
#ifdef RT_REGROUP_STATS
        rg_ticks[rg_stage] += wall_clock64() - rg_t0;
#endif
    }
    hand_over_counts();
    if (lane == 0) {
#ifdef RT_REGROUP_STATS
        for (uint32_t i = 0; i < RG_STATS; ++i)
            atomicAdd(&g_rg_stats[i], rg_turns[i]), atomicAdd(&g_rg_stats[RG_STATS + i], rg_lanes[i]), atomicAdd(&g_rg_stats[2 * RG_STATS + i], rg_ticks[i]);
#endif
    }
}

// ------------------------------------------------------------------ persistent path kernel, regrouping behind the wide BVH
// k_paths behind a tree ties a path to a lane: a lane whose traversal has ended waits for enough
// others before it is shaded (traversal lengths are heavy-tailed), and the shading code runs for
// the lanes that happen to need each branch.  On the 1 M-primitive scene that left 38 % of the lanes
// active, in the vector units and in the memory pipeline alike.  Here, as in k_paths_regroup, every
// WAVE owns a pool of POOL_SLOTS paths in LDS and one list of slot numbers per kind of pending work;
// the difference is the closest hit, which takes a different number of steps for every ray: a lane
// takes a ray from the ray list, keeps its traversal (cur, stack height, closest hit so far) in
// registers across turns, and hands the hit back to the pool when it is done -- then takes the
// next ray.  A turn of the wave is one of
//   * a traversal round for the lanes that hold a ray (trav4_round), while enough of them do;
//   * harvest and refill: lanes whose traversal has ended write the hit into their slot, which joins
//     the list of its material's kind; idle lanes take new rays from the ray list;
//   * a shading turn for up to 64 slots of one list, exactly k_paths_regroup's.
// LDS: the pool and POOL_STACK stack entries per lane; what a traversal pushes beyond that goes to a
// per-lane column in global memory (TravMem, SPILL) -- near-first order keeps few siblings pending.
// Every random number is keyed by (pixel, stratum, bounce), so the image equals k_paths'.
// Exit: a wave leaves when the sample counter has run out, no lane holds a ray and every list but
// the free one is empty; every traversal ends (the tree is finite) and every path within max_depth.
#ifndef RT_POOL_SLOTS
#define RT_POOL_SLOTS 120
#endif
#ifndef RT_POOL_STACK
#define RT_POOL_STACK 12
#endif
#ifndef RT_POOL_HARVEST
#define RT_POOL_HARVEST 16
#endif
#ifndef RT_POOL_STARVE
#define RT_POOL_STARVE 56
#endif

constexpr int POOL_SLOTS = RT_POOL_SLOTS;       // paths per wave
constexpr uint32_t POOL_STACK = RT_POOL_STACK;  // LDS stack entries per lane
constexpr uint32_t POOL_HARVEST = RT_POOL_HARVEST; // ended traversals that make the wave stop for a harvest
constexpr uint32_t POOL_STARVE = RT_POOL_STARVE;   // fewer traversing lanes than this: shade a list even if it is not full
struct TravPool {
    float4 ra[POOL_SLOTS], rb[POOL_SLOTS], sa[POOL_SLOTS];
    uint32_t sb[POOL_SLOTS];
    uint8_t list[RG_LISTS][POOL_SLOTS];
};
static_assert(POOL_SLOTS <= 256 && POOL_SLOTS >= WAVE && POOL_SLOTS % 8 == 0, "slot numbers are bytes; the struct stays 16-byte aligned");
// dynamic LDS of a block: the pools, then [POOL_STACK][WAVE] stack words per wave, then the staged nodes
constexpr size_t POOL_LDS_FIXED = (BLOCK / WAVE) * (sizeof(TravPool) + size_t(POOL_STACK) * WAVE * sizeof(uint32_t));

template <int MODE, bool QUANT, bool EXTRAS = true, bool PLAIN = false>
__global__ void __launch_bounds__(BLOCK, 4) k_paths_pool(const DScene *scp, const DCamera *camp, Film film, uint32_t k0, uint32_t k1,
                                                        Ctrl *ctrl, uint32_t has_media, uint32_t grab, uint32_t *spill,
                                                        uint32_t lds_nodes) {
    SceneRef sc = *as_uniform(scp);
    CamRef cam = *as_uniform(camp);
    extern __shared__ uint32_t s_dyn[];
    TravPool *pools = reinterpret_cast<TravPool *>(s_dyn);
    uint32_t *stacks = s_dyn + (BLOCK / WAVE) * sizeof(TravPool) / sizeof(uint32_t);
    float4 *l_nodes = reinterpret_cast<float4 *>(stacks + (BLOCK / WAVE) * POOL_STACK * WAVE);
    constexpr uint32_t NODE_BYTES = QUANT ? 64u : 128u;
    { // the only barrier of the kernel: the top of the tree, before any wave takes its first turn
        const uint32_t n = lds_nodes * (NODE_BYTES / 16u);
        for (uint32_t i = threadIdx.x; i < n; i += BLOCK) l_nodes[i] = sc.nodes4[i];
        __syncthreads();
    }
    const uint32_t wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    TravPool &pool = pools[wave];
    const TravMem mem{(lds_u32 *)(stacks + wave * POOL_STACK * WAVE + lane), reinterpret_cast<const char *>(l_nodes), lds_nodes * NODE_BYTES,
                      POOL_STACK, (global_u32 *)(spill + (size_t(blockIdx.x) * BLOCK + threadIdx.x)), gridDim.x * uint32_t(BLOCK)};
    for (uint32_t i = lane; i < uint32_t(POOL_SLOTS); i += WAVE) pool.list[RG_FREE][i] = uint8_t(i);
    uint32_t n_free = POOL_SLOTS, n_ray = 0, n_diffuse = 0, n_metal = 0, n_glass = 0, n_end = 0; // list lengths, wave-uniform
    const unsigned long long total = ctrl->total_samples;
    unsigned long long next = 0, end = 0, seen = 0;
    const float inv_share = 1.0f / (4.0f * float(gridDim.x * (BLOCK / WAVE)));
    uint32_t next_pixel = 0, next_stratum = 0;
    unsigned long long n_seg = 0, n_cam = 0;
    bool exhausted = false;
    // this lane's traversal
    bool busy = false; // holds a ray (its slot is in no list)
    uint32_t slot = 0, cur = RT_REF_NONE, sp = 0;
    float4 ra = {}, rb = {};
    RayCtx rc{k0, k1, 0u, 0u, 0u};
    Best best;
    best.t = RT_INF, best.ref = RT_REF_NONE, best.u = best.v = 0.f;
#ifdef RT_REGROUP_STATS
    unsigned long long rg_turns[RG_STATS] = {}, rg_lanes[RG_STATS] = {}, rg_ticks[RG_STATS] = {};
    unsigned long long rg_t0 = wall_clock64();
#define RT_POOL_TICK(stage) { const unsigned long long now_ = wall_clock64(); rg_ticks[stage] += now_ - rg_t0; rg_t0 = now_; }
#else
#define RT_POOL_TICK(stage) ((void)0)
#endif

    auto push = [&](uint32_t which, uint32_t &n, bool yes, uint32_t sl) {
        const unsigned long long mask = __ballot(yes);
        if (mask == 0) return; // uniform
        const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
        if (yes) pool.list[which][n + prefix] = uint8_t(sl);
        n += uint32_t(__popcll(mask));
    };
    // up to `want` new samples (see k_paths_regroup): lanes below the returned count make their
    // camera ray in `sl`, which joins the rays to trace
    auto new_paths = [&](uint32_t want, uint32_t sl) -> uint32_t {
        if (next == end && !exhausted) {
            const float left = float(total > seen ? total - seen : 0ull);
            uint32_t take = uint32_t(fminf(left * inv_share, float(grab)));
            take = take < uint32_t(WAVE) ? uint32_t(WAVE) : (take & ~uint32_t(WAVE - 1));
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(&ctrl->next_sample, static_cast<unsigned long long>(take));
            b = wave_first(b);
            seen = b;
            next = b < total ? b : total;
            end = b + take < total ? b + take : total;
            exhausted = next == end;
            next_pixel = uint32_t(next % cam.npix);
            next_stratum = cam.stratum_begin + uint32_t(next / cam.npix);
        }
        const uint32_t m = uint32_t(end - next < want ? end - next : want);
        if (m == 0) return 0;
        if (lane < m) {
            uint32_t px = next_pixel + lane, st = next_stratum;
            while (px >= cam.npix) px -= cam.npix, ++st;
            Rng rng;
            rng.begin(k0, k1, px, st, STREAM_CAMERA, 0);
            V3 o, d;
            float time;
            camera_ray(cam, px, st, rng, o, d, time);
            pool.ra[sl] = make_float4(o.x, o.y, o.z, time);
            pool.rb[sl] = make_float4(d.x, d.y, d.z, bitsf(RT_REF_NONE));
            pool.sa[sl] = make_float4(1.f, 1.f, 1.f, bitsf(px));
            pool.sb[sl] = st;
            pool.list[RG_RAY][n_ray + lane] = uint8_t(sl);
        }
        n_ray += m;
        next += m;
        n_cam += m;
        next_pixel += m;
        while (next_pixel >= cam.npix) next_pixel -= cam.npix, ++next_stratum;
        return m;
    };

    for (;;) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const bool traversing = busy && cur != RT_REF_NONE;
        const uint32_t n_trav = uint32_t(__popcll(__ballot(traversing)));
        const uint32_t n_done = uint32_t(__popcll(__ballot(busy && cur == RT_REF_NONE)));
        // rays waiting and lanes free to take them, or work that would make rays
        const uint32_t n_lanes_free = uint32_t(WAVE) - n_trav;
        const bool can_refill = n_ray != 0 && n_lanes_free >= POOL_HARVEST;
        const uint32_t fullest = max(max(n_diffuse, n_glass), max(n_metal, n_end));
        const bool can_shade = fullest >= uint32_t(WAVE) || (fullest != 0 && n_trav < POOL_STARVE && n_ray == 0) ||
                               (!exhausted && n_free != 0 && n_trav < POOL_STARVE && n_ray == 0);
        if (n_trav != 0 && n_done < POOL_HARVEST && !can_refill && !can_shade) {
            // 1. a traversal round for the lanes that hold a ray
            uint32_t rounds = 0;
            if (traversing) {
                // One round per call in the kernel compiled for any scene: running them here until the pool has
                // something else to do keeps 16 more scalar registers live through the rounds, and what that
                // kernel then spills to vector lanes costs more than the per-ray constants made again for every
                // round (measured: -12 %).  The PLAIN kernel has the registers (+1 % on the 1 M-primitive scene):
                // rounds until the conditions of the loop's head, with the lanes that finish inside the call
                // counted in, say that the pool has something else for the wave to do.
                auto more = [&](uint32_t nt, uint32_t nf) {
                    if (!PLAIN) return false;
                    const bool refill = n_ray != 0 && uint32_t(WAVE) - nt >= POOL_HARVEST;
                    const bool shade = fullest >= uint32_t(WAVE) || ((fullest != 0 || (!exhausted && n_free != 0)) && nt < POOL_STARVE && n_ray == 0);
                    return nt != 0 && n_done + nf < POOL_HARVEST && !refill && !shade;
                };
                if (lds_nodes) // uniform
                    rounds = trav4_rounds<true, QUANT, true, WAVE, EXTRAS, PLAIN>(sc, mk(ra), mk(rb), ra.w, 0.001f, fbits(rb.w), rc, mem, cur, sp, best, more);
                else
                    rounds = trav4_rounds<false, QUANT, true, WAVE, EXTRAS, PLAIN>(sc, mk(ra), mk(rb), ra.w, 0.001f, fbits(rb.w), rc, mem, cur, sp, best, more);
            }
            (void)rounds;
            RT_RG_COUNT(RG_RAY, n_trav);
            RT_POOL_TICK(RG_RAY);
            continue;
        }
        // 2. harvest: ended traversals hand their hit to the pool (camera.rs:444 answered)
        if (n_done != 0) {
            const bool done = busy && cur == RT_REF_NONE;
            uint32_t kind = RG_LISTS;
            if (done) {
                if (EXTRAS && sc.n_post) test_post(sc, mk(ra), mk(rb), ra.w, 0.001f, fbits(rb.w), rc, best); // uniform
                const V3 p = fma3(mk(rb), best.t, mk(ra)); // as shade_emit forms it
                pool.ra[slot] = make_float4(p.x, p.y, p.z, ra.w);
                pool.rb[slot].w = bitsf(best.ref); // the direction is still there
                // (a quad's (alpha, beta) are not kept: shade_emit recomputes them from the hit point
                // for the materials that look at them)
                // the hit's leaf said what its material asks for (a miss ends the path like a light)
                kind = best.cls == RT4_CLASS_END ? RG_END : (best.cls == RT4_CLASS_METAL ? RG_METAL : (best.cls == RT4_CLASS_GLASS ? RG_GLASS : RG_DIFFUSE));
                busy = false;
            }
            push(RG_DIFFUSE, n_diffuse, kind == RG_DIFFUSE, slot);
            push(RG_END, n_end, kind == RG_END, slot);
            push(RG_GLASS, n_glass, kind == RG_GLASS, slot);
            push(RG_METAL, n_metal, kind == RG_METAL, slot);
            RT_RG_COUNT(RG_LISTS, n_done);
            RT_POOL_TICK(RG_LISTS);
        }
        // 3. shading turns: a full list, or whatever there is when the traversal is short of rays
        {
            uint32_t stage = RG_LISTS, most = 0;
            auto offer = [&](uint32_t which, uint32_t n) {
                const uint32_t c = min(n, uint32_t(WAVE));
                if (c > most) most = c, stage = which;
            };
            offer(RG_DIFFUSE, n_diffuse);
            offer(RG_GLASS, n_glass);
            offer(RG_METAL, n_metal);
            offer(RG_END, n_end);
            if (!exhausted) offer(RG_FREE, n_free);
            const uint32_t n_hold = uint32_t(__popcll(__ballot(busy)));
            const bool starving = n_ray == 0 && n_hold < POOL_STARVE;
            if (most != 0 && (most >= uint32_t(WAVE) || starving)) {
                if (stage == RG_FREE) {
                    const uint32_t sl = lane < most ? pool.list[RG_FREE][n_free - 1 - lane] : 0u;
                    const uint32_t m = new_paths(most, sl);
                    n_free -= m;
                } else if (stage == RG_END) {
                    const uint32_t m = most;
                    n_end -= m;
                    const bool active = lane < m;
                    uint32_t sl = 0;
                    if (active) {
                        sl = pool.list[RG_END][n_end + lane];
                        const float4 srb = pool.rb[sl];
                        if (fbits(srb.w) != RG_REF_NOTHING) {
                            Surface sf; // ra holds the hit point: t = 0 from there
                            (void)shade_emit<true, false, true, PLAIN>(sc, cam, film, pool.ra[sl], srb, pool.sa[sl], make_float4(0.f, srb.w, 0.f, 0.f), true, sf);
                        }
                    }
                    const uint32_t taken = new_paths(m, sl);
                    push(RG_FREE, n_free, active && lane >= taken, sl);
                } else {
                    const uint32_t m = most;
                    auto scatter_turn = [&](auto only, uint32_t &n_list) {
                        constexpr uint32_t ONLY = decltype(only)::value;
                        n_list -= m;
                        const bool active = lane < m;
                        uint32_t sl = 0;
                        bool alive = false;
                        if (active) {
                            sl = pool.list[stage][n_list + lane];
                            float4 sra = pool.ra[sl], srb = pool.rb[sl], ssa = pool.sa[sl];
                            const uint32_t packed = pool.sb[sl];
                            uint2 ssb = make_uint2(packed & 0xFFFFFFu, packed >> 24);
                            Surface sf;
                            (void)shade_emit<false, false, true, PLAIN>(sc, cam, film, sra, srb, ssa, make_float4(0.f, srb.w, 0.f, 0.f), true, sf);
                            Rng rng;
                            rng.begin(k0, k1, fbits(ssa.w), ssb.x, STREAM_SHADE, ssb.y);
                            alive = shade_scatter<MODE, ONLY, false, PLAIN>(sc, cam, sra, srb, ssa, ssb, sf, rng, ctrl);
                            if (alive) {
                                pool.ra[sl] = sra;
                                pool.rb[sl] = srb;
                                pool.sa[sl] = ssa;
                                pool.sb[sl] = ssb.x | (ssb.y << 24);
                            } else {
                                pool.rb[sl].w = bitsf(RG_REF_NOTHING);
                            }
                        }
                        push(RG_RAY, n_ray, alive, sl);
                        push(RG_END, n_end, active && !alive, sl);
                    };
                    if (stage == RG_DIFFUSE) scatter_turn(std::integral_constant<uint32_t, RT_MAT_LAMBERTIAN>{}, n_diffuse);
                    else if (stage == RG_GLASS) scatter_turn(std::integral_constant<uint32_t, RT_MAT_DIELECTRIC>{}, n_glass);
                    else scatter_turn(std::integral_constant<uint32_t, RT_MAT_METAL>{}, n_metal);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                RT_RG_COUNT(stage, most);
                RT_POOL_TICK(stage);
            } else if (most == 0 && n_hold == 0 && n_ray == 0) {
                break; // nothing left anywhere (the free list does not count once the samples have run out)
            }
        }
        // 4. refill: lanes without a ray take the rays waiting
        if (n_ray != 0) {
            const unsigned long long idle = __ballot(!busy);
            const uint32_t n_idle = uint32_t(__popcll(idle));
            const uint32_t m = min(n_idle, n_ray);
            if (m != 0) {
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle), 0u));
                if (!busy && prefix < m) {
                    slot = pool.list[RG_RAY][n_ray - 1 - prefix];
                    ra = pool.ra[slot];
                    rb = pool.rb[slot];
                    if (has_media) { // a medium keys its free-flight draw by the sample and the bounce
                        const uint32_t sb = pool.sb[slot];
                        rc.pixel = fbits(pool.sa[slot].w), rc.sample = sb & 0xFFFFFFu, rc.bounce = sb >> 24;
                    }
                    trav4_begin(RT_INF, cur, sp, best);
                    busy = true;
                }
                n_ray -= m;
                n_seg += m;
                RT_RG_COUNT(RG_LISTS + 1, m);
                RT_POOL_TICK(RG_LISTS + 1);
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&ctrl->segments, n_seg);
        atomicAdd(&ctrl->camera_rays, n_cam);
#ifdef RT_REGROUP_STATS
        for (uint32_t i = 0; i < RG_STATS; ++i)
            atomicAdd(&g_rg_stats[i], rg_turns[i]), atomicAdd(&g_rg_stats[RG_STATS + i], rg_lanes[i]), atomicAdd(&g_rg_stats[2 * RG_STATS + i], rg_ticks[i]);
#endif
    }
}

// ------------------------------------------------------------------ hooks
template <int ACCEL>
__global__ void __launch_bounds__(BLOCK) k_intersect_batch(const DScene *scp, uint32_t n, const float *o3, const float *d3,
                                                            const float *time, const uint32_t *origin_ref, float tmin, float tmax,
                                                            int *hit,
                                                            int *prim, float *t, float *u, float *v) {
    SceneRef sc = *as_uniform(scp);
    RT_ACCEL_LDS(ACCEL)
    if (accel_is_wide(ACCEL)) stage_nodes4<ACCEL == ACCEL_BVH4Q>(sc, s_dyn);
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    V3 o = {o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]}, d = {d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]};
    float tm = time ? time[i] : 0.f;
    RayCtx rc{0u, 0u, i, 0u, 0u};
    const uint32_t from = origin_ref ? origin_ref[i] : RT_REF_NONE;
    Best best;
    closest_hit<ACCEL, false>(sc, o, d, tm, tmin, tmax, from, rc, s_stack, s_nodes, s_dyn, best);
    int pid = -1;
    float uu = best.u, vv = best.v;
    if (best.ref != RT_REF_NONE) {
        uint32_t kind = best.ref >> RT_REF_KIND_SHIFT, idx = best.ref & RT_REF_INDEX_MASK;
        if (kind == RT_KIND_QUAD) {
            pid = int(fbits(sc.quads[6 * idx + 4].w));
            if (ACCEL == ACCEL_FLAT) quad_uv(sc.quads + 6 * idx, fma3(d, best.t, o), uu, vv);
            else if (uu != uu) quad_uv_ray(sc.quads + 6 * idx, o, d, best.t, uu, vv);
        } else if (kind == RT_KIND_SPHERE) {
            const float4 *S = sc.spheres + 3 * idx;
            float4 s0 = S[0], s1 = S[1], s2 = S[2];
            pid = int(fbits(s2.x));
            V3 center = fma3(mk(s1), tm, mk(s0));
            V3 nn = (fma3(d, best.t, o) - center) * (1.0f / s0.w);
            sphere_uv(nn, s2.z, uu, vv);
        } else {
            pid = int(sc.media[idx].prim_id);
            uu = vv = __builtin_nanf("");
        }
    }
    hit[i] = best.ref != RT_REF_NONE;
    prim[i] = pid;
    t[i] = best.ref != RT_REF_NONE ? best.t : RT_INF;
    u[i] = uu;
    v[i] = vv;
}

__global__ void k_camera_rays(const DCamera *camp, uint32_t k0, uint32_t k1, uint32_t n, const uint32_t *pixel,
                              const uint32_t *stratum, float *o3, float *d3, float *time) {
    CamRef cam = *as_uniform(camp);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o, d;
    float tm;
    camera_ray(cam, pixel[i], stratum[i], k0, k1, o, d, tm);
    o3[3 * i] = o.x, o3[3 * i + 1] = o.y, o3[3 * i + 2] = o.z;
    d3[3 * i] = d.x, d3[3 * i + 1] = d.y, d3[3 * i + 2] = d.z;
    time[i] = tm;
}

__global__ void k_eval_texture(const DScene *scp, uint32_t tex, uint32_t n, const float *uvp, float *rgb) {
    SceneRef sc = *as_uniform(scp);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 c = tex_value(sc, tex, uvp[5 * i], uvp[5 * i + 1], V3{uvp[5 * i + 2], uvp[5 * i + 3], uvp[5 * i + 4]});
    rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
}

__global__ void k_philox(uint32_t n, const uint32_t *ctr, uint32_t k0, uint32_t k1, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t o[4];
    philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], k0, k1, o);
    for (int k = 0; k < 4; ++k) out[4 * i + k] = o[k];
}

// film planes -> interleaved rgb * scale
__global__ void k_film_pack(const float *r, const float *g, const float *b, uint32_t npix, float scale, float *rgb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    rgb[3 * i] = r[i] * scale;
    rgb[3 * i + 1] = g[i] * scale;
    rgb[3 * i + 2] = b[i] * scale;
}

// interleaved rgb sums -> film planes (restoring a saved film)
__global__ void k_film_unpack(const float *rgb, uint32_t npix, float *r, float *g, float *b) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    r[i] = rgb[3 * i];
    g[i] = rgb[3 * i + 1];
    b[i] = rgb[3 * i + 2];
}

// Output stage: film sums -> 8-bit RGB with the reference's rule: value = sum * scale
// (camera.rs:397), optional gamma 2 = sqrt, 0 for values <= 0 (color.rs:209-215), clamp to
// [0, 1], * 255, truncate (color.rs:148-155; NaN -> 0 as a Rust `as u8` cast does).  Evaluated
// in f64 like the reference, so the bytes equal the reference rule applied to the same sums.
// One thread packs four pixels = 12 bytes = three dword stores.
__device__ __forceinline__ uint32_t quantize_channel(float sum, double scale, bool gamma) {
    double c = double(sum) * scale;
    if (gamma) c = c > 0.0 ? sqrt(c) : 0.0;
    c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c); // Interval::clamp (interval.rs:151-153); NaN passes through
    c *= 255.0;
    return c >= 0.0 ? uint32_t(c) : 0u;       // NaN fails the comparison -> 0
}
__global__ void k_film_quantize(const float *r, const float *g, const float *b, uint32_t npix, double scale,
                                uint32_t gamma, uint32_t *out /* ceil(npix/4) * 3 dwords */) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; // group of four pixels
    if (q * 4u >= npix) return;
    uint32_t bytes[12];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = q * 4u + k;
        const bool in = i < npix;
        bytes[3 * k] = in ? quantize_channel(r[i], scale, gamma != 0) : 0u;
        bytes[3 * k + 1] = in ? quantize_channel(g[i], scale, gamma != 0) : 0u;
        bytes[3 * k + 2] = in ? quantize_channel(b[i], scale, gamma != 0) : 0u;
    }
#pragma unroll
    for (uint32_t w = 0; w < 3; ++w)
        out[q * 3u + w] = bytes[4 * w] | (bytes[4 * w + 1] << 8) | (bytes[4 * w + 2] << 16) | (bytes[4 * w + 3] << 24);
}

} // namespace rt
