// oracle.cpp -- CPU oracle: f64 restatement of the reference render path.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h).  Structure follows the reference on purpose:
// virtual Hittable / Material / Texture / PDF objects, recursive ray_color, one thread.
// Citations are file:line in the reference tree (src/...).
//
// Parity pin: reference KATs only (sphere uv, Interval doc-tests); the reference itself cannot
// be built here.  Estimator mode BOOK3, sphere/list lights and Perlin noise have no
// counterpart in the reference and are PARITY UNPINNED.

#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <optional>
#include <random>
#include <string>
#include <vector>

namespace {

constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double INF = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------- math (math/vec.rs)
struct V3 {
    double x = 0, y = 0, z = 0;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator-(V3 a, V3 b) { return a + (-b); }                         // vec.rs:408-414
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; } // vec.rs:456-467
inline V3 operator*(double s, V3 a) { return a * s; }
inline V3 operator/(V3 a, double s) { return a * (1.0 / s); } // vec.rs:484-490 (reciprocal multiply)
// vec.rs:122-124: dot is a nested mul_add
inline double dot(V3 a, V3 b) { return std::fma(a.x, b.x, std::fma(a.y, b.y, a.z * b.z)); }
// vec.rs:163-170 / 299-306: cross via mul_add
inline V3 cross(V3 a, V3 b) {
    return {std::fma(a.y, b.z, -(a.z * b.y)), std::fma(a.z, b.x, -(a.x * b.z)),
            std::fma(a.x, b.y, -(a.y * b.x))};
}
inline double len_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; } // vec.rs:111-113
inline double len(V3 a) { return std::sqrt(len_squared(a)); }
inline V3 unit(V3 a) { return a / len(a); } // vec.rs:259-267
// vec.rs:156-159
inline V3 reflect(V3 v, V3 n) {
    double mul = 2.0 * dot(v, n);
    return v - n * mul;
}
// vec.rs:289-295
inline V3 refract(V3 uv, V3 n, double ratio) {
    double cos_theta = std::fmin(dot(-uv, n), 1.0);
    V3 perp = ratio * (uv + (cos_theta * n));
    V3 par = -std::sqrt(std::fabs(1.0 - len_squared(perp))) * n;
    return perp + par;
}

struct Color {
    double r = 0, g = 0, b = 0;
};
inline Color cadd(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; } // color.rs:179-185
inline Color cmul(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; } // color.rs:187-193
inline Color cscale(Color a, double s) { return {a.r * s, a.g * s, a.b * s}; }    // color.rs:173-177

// color.rs:46-64
Color color_hex(uint32_t color) {
    uint32_t r, g, b;
    if (color <= 0xfff) {
        r = (color & 0xf00) >> 8;
        g = ((color << 4) & 0xf00) >> 8;
        b = ((color << 8) & 0xf00) >> 8;
        r *= 16;
        g *= 16;
        b *= 16;
    } else {
        r = (color & 0xff0000) >> 16;
        g = ((color << 8) & 0xff0000) >> 16;
        b = ((color << 16) & 0xff0000) >> 16;
    }
    return {double(r) / 255.0, double(g) / 255.0, double(b) / 255.0};
}

// ---------------------------------------------------------------- Interval (math/interval.rs)
struct Interval {
    double start, end;
    static Interval positive(double a, double b) { return a <= b ? Interval{a, b} : Interval{b, a}; } // :23-29
    double size() const { return end - start; }                                                      // :81-83
    bool contains(double x) const { return start <= x && x <= end; }                                  // :86-88
    bool surrounds(double x) const { return start < x && x < end; }                                   // :114-116
    std::optional<Interval> overlap(const Interval &o) const {                                        // :44-54
        Interval r{std::fmax(start, o.start), std::fmin(end, o.end)};
        if (r.size() <= 0.0) return std::nullopt;
        return r;
    }
    static Interval surrounding(const Interval &a, const Interval &b) { // :66-71
        return positive(std::fmin(a.start, b.start), std::fmax(a.end, b.end));
    }
    Interval expand(double delta) const { return {start - delta / 2.0, end + delta / 2.0}; } // :151-153
    double clamp(double x) const { return x < start ? start : (x > end ? end : x); }          // :130-138
    static Interval empty() { return {INF, -INF}; }
    static Interval universe() { return {-INF, INF}; }
};

// ---------------------------------------------------------------- RNG
// The reference draws from rand 0.9 ThreadRng (ChaCha12, OS seeded: Cargo.lock:371-383), so no
// stream of ours can match it bit for bit.  The oracle's default stream is the counter-based
// one the device uses, so that CPU and GPU renders see the same uniforms.
inline void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = uint64_t(0xD2511F53u) * c0;
        uint64_t p1 = uint64_t(0xCD9E8D57u) * c2;
        uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = uint32_t(p1);
        uint32_t n2 = uint32_t(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = uint32_t(p0);
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

enum : uint32_t { STREAM_CAMERA = 0, STREAM_SHADE = 1 };

struct Sampler {
    int kind = ORC_RNG_PHILOX24;
    uint32_t key[2] = {0, 0};
    uint32_t pixel = 0, sample = 0;
    uint32_t c2 = 0, block = 0;
    uint32_t buf[4];
    int pos = 4;
    std::mt19937_64 mt;

    void begin(uint32_t stream, uint32_t bounce) {
        c2 = (bounce << 8) | stream;
        block = 0;
        pos = 4;
    }
    uint32_t next_u32() {
        if (pos == 4) {
            uint32_t c[4] = {pixel, sample, c2, block++};
            philox4x32_10(c, key, buf);
            pos = 0;
        }
        return buf[pos++];
    }
    static double f24(uint32_t u) { return double(u >> 8) * (1.0 / 16777216.0); }
    static double f53(uint32_t hi, uint32_t lo) {
        return double(((uint64_t(hi) << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
    }
    double next() {
        if (kind == ORC_RNG_MT64) return double(mt() >> 11) * (1.0 / 9007199254740992.0);
        if (kind == ORC_RNG_PHILOX53) {
            uint32_t a = next_u32();
            uint32_t b = next_u32();
            return f53(a, b);
        }
        return f24(next_u32());
    }
    // one draw per (segment, medium), independent of traversal order: Philox2x32-10 (Random123),
    // counter = (pixel, stratum), key = seed, bounce and the medium's primitive id folded into a word
    static void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t k, uint32_t out[2]) {
        for (int r = 0; r < 10; ++r) {
            if (r) k += 0x9E3779B9u;
            const uint64_t p = uint64_t(0xD256D193u) * c0;
            const uint32_t n0 = uint32_t(p >> 32) ^ k ^ c1;
            c1 = uint32_t(p);
            c0 = n0;
        }
        out[0] = c0, out[1] = c1;
    }
    double medium_draw(uint32_t bounce, uint32_t medium_id) {
        if (kind == ORC_RNG_MT64) return next();
        const uint32_t k = key[0] + 0x9E3779B9u * (key[1] ^ (bounce * 0x85EBCA6Bu) ^ (medium_id * 0xC2B2AE35u) ^ 0x27D4EB2Fu);
        uint32_t o[2];
        philox2x32_10(pixel, sample, k, o);
        return kind == ORC_RNG_PHILOX53 ? f53(o[0], o[1]) : f24(o[0]);
    }
    double range_pm1() { return 2.0 * next() - 1.0; } // Uniform(-1.0..=1.0), vec.rs:213
};

struct Ctx {
    Sampler rng;
    uint32_t bounce = 0;
    int mode = ORC_MODE_REF_EXACT;
    bool sanitize = false;
    int exclude_prim = -1;
    int error = 0; // 1: material not renderable in this mode (reference panics)
    uint64_t cnt[ORC_NCOUNTERS] = {0};
    // orc_render_flagged: the path in flight has scattered off a sphere smaller than flag_radius
    double flag_radius = 0;
    bool path_flag = false;
};

// vec.rs:211-223
V3 random_in_unit_sphere(Ctx &c) {
    for (;;) {
        V3 p;
        p.x = c.rng.range_pm1();
        p.y = c.rng.range_pm1();
        p.z = c.rng.range_pm1();
        double l2 = len_squared(p);
        if (1.0e-160 < l2 && l2 < 1.0) return p;
    }
}
// vec.rs:246-256
V3 cosine_from(double r1, double r2) {
    double phi = 2.0 * PI * r1;
    double x = std::cos(phi) * std::sqrt(r2);
    double y = std::sin(phi) * std::sqrt(r2);
    double z = std::sqrt(1.0 - r2);
    return {x, y, z};
}
V3 random_on_sphere_cosine(Ctx &c) {
    double r1 = c.rng.next();
    double r2 = c.rng.next();
    return cosine_from(r1, r2);
}

// ---------------------------------------------------------------- Ray (math/ray.rs)
struct Ray {
    V3 o, d;
    double time = 0;
    V3 at(double t) const { return o + d * t; } // ray.rs:45-47
};

// ---------------------------------------------------------------- ONB (onb.rs:13-45)
struct ONB {
    V3 u, v, w;
    explicit ONB(V3 n) {
        w = unit(n);
        V3 a = std::fabs(w.x) > 0.9 ? V3{0, 1, 0} : V3{1, 0, 0};
        a = unit(a);
        v = unit(cross(w, a));
        u = cross(w, v);
    }
    V3 transform(V3 p) const { return (p.x * u) + (p.y * v) + (p.z * w); }
};

// ---------------------------------------------------------------- BoundingBox3 (boundingbox.rs)
struct BBox {
    Interval x, y, z;
    const Interval &ax(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    static BBox make(Interval x, Interval y, Interval z) { // :14-30
        const double DELTA = 0.0001;
        auto expand = [&](Interval i) { return i.size() < DELTA ? i.expand(DELTA) : i; };
        return {expand(x), expand(y), expand(z)};
    }
    static BBox empty() { return {Interval::empty(), Interval::empty(), Interval::empty()}; }
    static BBox bounded_by(V3 a, V3 b) { // :51-56
        return make(Interval::positive(a.x, b.x), Interval::positive(a.y, b.y), Interval::positive(a.z, b.z));
    }
    static BBox extending(const BBox &a, const BBox &b) { // :58-63
        return make(Interval::surrounding(a.x, b.x), Interval::surrounding(a.y, b.y),
                    Interval::surrounding(a.z, b.z));
    }
    int longest_axis() const { // :75-94 (strict >, ties -> earlier axis, empty -> X)
        int axis = 0;
        double best = 0.0;
        for (int i = 0; i < 3; ++i) {
            double s = ax(i).size();
            if (s > best) {
                best = s;
                axis = i;
            }
        }
        return axis;
    }
    BBox shifted(V3 off) const { // :140-146
        return make({x.start + off.x, x.end + off.x}, {y.start + off.y, y.end + off.y},
                    {z.start + off.z, z.end + off.z});
    }
    bool hit(const Ray &r, Interval ray_t) const { // :96-113
        for (int a = 0; a < 3; ++a) {
            const Interval &ai = ax(a);
            double adinv = 1.0 / r.d[a];
            double t0 = (ai.start - r.o[a]) * adinv;
            double t1 = (ai.end - r.o[a]) * adinv;
            Interval ti = Interval::positive(t0, t1);
            auto ov = ti.overlap(ray_t);
            if (!ov) return false;
            ray_t = *ov;
        }
        return true;
    }
};

// ---------------------------------------------------------------- Texture (texture.rs)
struct Texture {
    virtual ~Texture() = default;
    virtual Color value(double u, double v, V3 p) const = 0;
};
struct SolidColor : Texture { // :18-30
    Color c;
    explicit SolidColor(Color c) : c(c) {}
    Color value(double, double, V3) const override { return c; }
};
struct Checkerboard : Texture { // :39-72
    double scale;
    std::shared_ptr<Texture> even, odd;
    Color value(double u, double v, V3 p) const override {
        double x = std::floor((1.0 / scale) * p.x);
        double y = std::floor((1.0 / scale) * p.y);
        double z = std::floor((1.0 / scale) * p.z);
        // `as i32` saturates; wrapping add would differ only beyond +-2^31 cells
        auto as_i32 = [](double f) -> int32_t {
            if (f != f) return 0;
            if (f >= 2147483647.0) return INT32_MAX;
            if (f <= -2147483648.0) return INT32_MIN;
            return int32_t(f);
        };
        int64_t s = int64_t(as_i32(x)) + as_i32(y) + as_i32(z);
        bool is_even = (s % 2) == 0; // negative odd sums give -1 -> odd, as in Rust
        return is_even ? even->value(u, v, p) : odd->value(u, v, p);
    }
};
struct ImageTexture : Texture { // :75-140
    std::vector<uint8_t> data;
    uint32_t w = 0, h = 0;
    Color value(double u, double v, V3) const override {
        // The reference asserts u,v in [0,1] and then indexes out of bounds at u=1 or v=0
        // (texture.rs:131-134).  The oracle clamps the texel index instead of panicking.
        u = std::fmin(std::fmax(u, 0.0), 1.0);
        v = std::fmin(std::fmax(v, 0.0), 1.0);
        v = 1.0 - v;
        uint32_t i = uint32_t(u * double(w));
        uint32_t j = uint32_t(v * double(h));
        if (i > w - 1) i = w - 1;
        if (j > h - 1) j = h - 1;
        size_t ind = (size_t(j) * w + i) * 3;
        return {double(data[ind]) / 255.0, double(data[ind + 1]) / 255.0, double(data[ind + 2]) / 255.0};
    }
};
// Not in the reference (no Perlin texture exists there).  "Ray Tracing: The Next Week" ch.5:
// gradient noise with hermite smoothing, 7-octave turbulence, marble phase on z.
struct NoiseTexture : Texture {
    double scale = 1;
    V3 ranvec[256];
    int px[256], py[256], pz[256];
    double noise(V3 p) const {
        double u = p.x - std::floor(p.x), v = p.y - std::floor(p.y), w = p.z - std::floor(p.z);
        int i = int(std::floor(p.x)), j = int(std::floor(p.y)), k = int(std::floor(p.z));
        double uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
        double accum = 0;
        for (int di = 0; di < 2; ++di)
            for (int dj = 0; dj < 2; ++dj)
                for (int dk = 0; dk < 2; ++dk) {
                    V3 c = ranvec[px[(i + di) & 255] ^ py[(j + dj) & 255] ^ pz[(k + dk) & 255]];
                    V3 weight{u - di, v - dj, w - dk};
                    accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) *
                             (dk * ww + (1 - dk) * (1 - ww)) * (c.x * weight.x + c.y * weight.y + c.z * weight.z);
                }
        return accum;
    }
    double turb(V3 p, int depth) const {
        double accum = 0, weight = 1;
        for (int i = 0; i < depth; ++i) {
            accum += weight * noise(p);
            weight *= 0.5;
            p = p * 2.0;
        }
        return std::fabs(accum);
    }
    Color value(double, double, V3 p) const override {
        double s = 0.5 * (1.0 + std::sin(scale * p.z + 10.0 * turb(p, 7)));
        return {s, s, s};
    }
};

// ---------------------------------------------------------------- HitRecord / Material
struct Material;
struct HitRecord { // hittable.rs:13-28
    V3 point, normal;
    const Material *material = nullptr;
    double u = 0, v = 0, t = 0;
    bool front_face = false;
    int prim = -1;
    double sphere_radius = 0; // > 0: the hit is on a sphere of that radius (orc_render_flagged only)
    // hittable.rs:59-79
    static HitRecord from_incoming_ray(const Ray &ray, V3 point, V3 normal, double t, double u, double v,
                                       const Material *m, int prim) {
        HitRecord h;
        h.front_face = dot(ray.d, normal) < 0.0;
        h.normal = h.front_face ? normal : -normal;
        h.point = point;
        h.t = t;
        h.u = u;
        h.v = v;
        h.material = m;
        h.prim = prim;
        return h;
    }
};

enum PdfKind { PDF_NONE = 0, PDF_COSINE = 1, PDF_SPHERE = 2 };

struct MaterialResult { // material.rs:8-12
    Color attenuation;
    Ray scattered;
    double pdf = 0;
    bool skip_pdf = false;   // BOOK3: specular, bypass the PDF machinery
    int pdf_kind = PDF_NONE; // BOOK3: which material PDF the mixture uses
};

struct Material {
    virtual ~Material() = default;
    virtual std::optional<MaterialResult> scatter(const Ray &in, const HitRecord &rec, Ctx &c) const = 0;
    virtual Color emitted(const Ray &, const HitRecord &, double, double, V3) const { return {}; } // :16-18
    virtual double scattering_pdf(const Ray &, const HitRecord &, const Ray &, Ctx &c) const {
        c.error = 1; // material.rs:27-29 unimplemented!()
        return 0.0;
    }
};

struct Lambertian : Material { // material.rs:33-63
    std::shared_ptr<Texture> tex;
    std::optional<MaterialResult> scatter(const Ray &in, const HitRecord &rec, Ctx &c) const override {
        MaterialResult r;
        r.attenuation = tex->value(rec.u, rec.v, rec.point);
        if (c.mode == ORC_MODE_BOOK3) {
            // the mixture draws the direction; the material only names its PDF
            r.pdf_kind = PDF_COSINE;
            return r;
        }
        ONB uvw(rec.normal);
        V3 dir = uvw.transform(random_on_sphere_cosine(c));
        r.scattered = Ray{rec.point, unit(dir), in.time};
        r.pdf = dot(uvw.w, r.scattered.d) / PI;
        r.pdf_kind = PDF_COSINE;
        return r;
    }
    double scattering_pdf(const Ray &, const HitRecord &rec, const Ray &sc, Ctx &) const override {
        double cos_theta = dot(rec.normal, unit(sc.d));
        return std::fmax(0.0, cos_theta / PI);
    }
};

struct DiffuseLight : Material { // material.rs:66-92
    std::shared_ptr<Texture> tex;
    std::optional<MaterialResult> scatter(const Ray &, const HitRecord &, Ctx &) const override {
        return std::nullopt;
    }
    Color emitted(const Ray &, const HitRecord &rec, double u, double v, V3 p) const override {
        return rec.front_face ? tex->value(u, v, p) : Color{};
    }
};

struct Metal : Material { // material.rs:95-132
    Color albedo;
    double fuzz = 0;
    std::optional<MaterialResult> scatter(const Ray &in, const HitRecord &rec, Ctx &c) const override {
        if (c.mode == ORC_MODE_REF_EXACT) {
            c.error = 1; // `pdf: todo!()` panics at material.rs:128
            return std::nullopt;
        }
        V3 reflected = reflect(in.d, rec.normal);
        reflected = unit(reflected) + (fuzz * random_in_unit_sphere(c));
        if (dot(reflected, rec.normal) < 0.0) return std::nullopt;
        MaterialResult r;
        r.attenuation = albedo;
        r.scattered = Ray{rec.point, reflected, in.time};
        r.skip_pdf = true;
        return r;
    }
};

// material.rs:146-151
double schlick(double cosine, double ri) {
    double r0 = (1.0 - ri) / (1.0 + ri);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * std::pow(1.0 - cosine, 5.0);
}

struct Dielectric : Material { // material.rs:135-183
    double ir = 1.5;
    std::optional<MaterialResult> scatter(const Ray &in, const HitRecord &rec, Ctx &c) const override {
        if (c.mode == ORC_MODE_REF_EXACT) {
            c.error = 1; // `pdf: todo!()` panics at material.rs:179
            return std::nullopt;
        }
        double ri = rec.front_face ? 1.0 / ir : ir;
        V3 direction = unit(in.d);
        double cos_theta = std::fmin(dot(-direction, rec.normal), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > 1.0;
        // `||` short-circuits: the draw happens only when refraction is possible
        bool will_reflect = cannot_refract || (schlick(cos_theta, ri) > c.rng.next());
        V3 out = will_reflect ? reflect(direction, rec.normal) : refract(direction, rec.normal, ri);
        MaterialResult r;
        r.attenuation = {1, 1, 1};
        r.scattered = Ray{rec.point, out, in.time};
        r.skip_pdf = true;
        return r;
    }
};

struct Isotropic : Material { // material.rs:186-213
    std::shared_ptr<Texture> tex;
    std::optional<MaterialResult> scatter(const Ray &in, const HitRecord &rec, Ctx &c) const override {
        MaterialResult r;
        r.attenuation = tex->value(rec.u, rec.v, rec.point);
        r.pdf = 1.0 / (4.0 * PI);
        r.pdf_kind = PDF_SPHERE;
        if (c.mode == ORC_MODE_BOOK3) return r;
        r.scattered = Ray{rec.point, random_in_unit_sphere(c), in.time};
        return r;
    }
    double scattering_pdf(const Ray &, const HitRecord &, const Ray &, Ctx &) const override {
        return 1.0 / (4.0 * PI);
    }
};

// ---------------------------------------------------------------- Hittable (hittable.rs)
struct Hittable {
    virtual ~Hittable() = default;
    virtual std::optional<HitRecord> hit(const Ray &ray, Interval ray_t, Ctx &c) const = 0;
    virtual const BBox *bounding_box() const = 0;
    virtual double pdf_value(V3, V3, Ctx &c) const { // :99-102 unimplemented!()
        c.error = 2;
        return 0.0;
    }
    virtual V3 random(V3, Ctx &c) const { // :104-107
        c.error = 2;
        return {1, 0, 0};
    }
};
using HPtr = std::shared_ptr<Hittable>;

// hittable.rs:150-162
void sphere_get_uv(V3 p, double &u, double &v) {
    double theta = std::acos(-p.y);
    double phi = std::atan2(-p.z, p.x) + PI;
    u = phi / (2.0 * PI);
    v = theta / PI;
}

struct Sphere : Hittable { // hittable.rs:117-209
    V3 c0, cv;
    double radius;
    const Material *mat;
    BBox bbox;
    int prim;
    Sphere(V3 c0_, V3 cv_, double r, const Material *m, int prim_) : c0(c0_), cv(cv_), radius(r), mat(m), prim(prim_) {
        V3 rad{r, r, r};
        BBox b0 = BBox::bounded_by(c0 - rad, c0 + rad);
        V3 c1 = c0 + cv * 1.0;
        BBox b1 = BBox::bounded_by(c1 - rad, c1 + rad);
        bbox = BBox::extending(b0, b1);
    }
    std::optional<HitRecord> hit(const Ray &ray, Interval ray_t, Ctx &c) const override {
        if (c.exclude_prim == prim) return std::nullopt;
        c.cnt[2]++;
        V3 center = c0 + cv * ray.time;
        V3 oc = center - ray.o;
        double a = len_squared(ray.d);
        double h = dot(ray.d, oc);
        double cc = len_squared(oc) - radius * radius;
        double disc = h * h - a * cc;
        if (disc < 0.0) return std::nullopt;
        double sqrtd = std::sqrt(disc);
        double root = (h - sqrtd) / a;
        if (!ray_t.surrounds(root)) {
            root = (h + sqrtd) / a;
            if (!ray_t.surrounds(root)) return std::nullopt;
        }
        c.cnt[3]++;
        V3 point = ray.at(root);
        V3 normal = (point - center) / radius;
        double u, v;
        sphere_get_uv(normal, u, v);
        HitRecord rec = HitRecord::from_incoming_ray(ray, point, normal, root, u, v, mat, prim);
        rec.sphere_radius = radius;
        return rec;
    }
    const BBox *bounding_box() const override { return &bbox; }
    // Not in the reference (only Parallelogram overrides these).  "The Rest of Your Life" ch.12.
    double pdf_value(V3 origin, V3 direction, Ctx &c) const override {
        Ray r{origin, direction, 0.0};
        int save = c.exclude_prim;
        c.exclude_prim = -1;
        auto h = hit(r, Interval{0.001, INF}, c);
        c.exclude_prim = save;
        if (!h) return 0.0;
        double dist2 = len_squared(c0 - origin);
        double cos_theta_max = std::sqrt(1.0 - radius * radius / dist2);
        double solid_angle = 2.0 * PI * (1.0 - cos_theta_max);
        return 1.0 / solid_angle;
    }
    V3 random(V3 origin, Ctx &c) const override {
        V3 direction = c0 - origin;
        double dist2 = len_squared(direction);
        ONB uvw(direction);
        double r1 = c.rng.next();
        double r2 = c.rng.next();
        double z = 1.0 + r2 * (std::sqrt(1.0 - radius * radius / dist2) - 1.0);
        double phi = 2.0 * PI * r1;
        double s = std::sqrt(1.0 - z * z);
        return uvw.transform(V3{std::cos(phi) * s, std::sin(phi) * s, z});
    }
};

struct HittableVec : Hittable { // hittable.rs:211-269
    std::vector<HPtr> objects;
    std::optional<BBox> bbox;
    void add(HPtr o) { // :238-244
        bbox = bbox ? BBox::extending(*bbox, *o->bounding_box()) : *o->bounding_box();
        objects.push_back(std::move(o));
    }
    std::optional<HitRecord> hit(const Ray &ray, Interval ray_t, Ctx &c) const override { // :252-264
        std::optional<HitRecord> closest;
        double closest_dist = ray_t.end;
        for (const auto &o : objects) {
            if (auto rec = o->hit(ray, Interval{ray_t.start, closest_dist}, c)) {
                closest_dist = rec->t;
                closest = rec;
            }
        }
        return closest;
    }
    const BBox *bounding_box() const override { return bbox ? &*bbox : nullptr; }
    // Not in the reference: light lists per "The Rest of Your Life" ch.12.3
    double pdf_value(V3 origin, V3 direction, Ctx &c) const override {
        double weight = 1.0 / double(objects.size());
        double sum = 0.0;
        for (const auto &o : objects) sum += weight * o->pdf_value(origin, direction, c);
        return sum;
    }
    V3 random(V3 origin, Ctx &c) const override {
        size_t n = objects.size();
        size_t i = size_t(c.rng.next() * double(n));
        if (i > n - 1) i = n - 1;
        return objects[i]->random(origin, c);
    }
};

struct Quad : Hittable { // Parallelogram / Triangle / Disc, hittable.rs:281-677
    int shape;
    V3 corner, u, v, w, normal;
    double d, area;
    const Material *mat;
    BBox bbox;
    int prim;
    Quad(int shape_, V3 q, V3 u_, V3 v_, const Material *m, int prim_)
        : shape(shape_), corner(q), u(u_), v(v_), mat(m), prim(prim_) {
        BBox d1 = BBox::bounded_by(corner, corner + u + v); // :296-299
        BBox d2 = BBox::bounded_by(corner + u, corner + v);
        bbox = BBox::extending(d1, d2);
        V3 n = cross(u, v);
        normal = unit(n);
        d = dot(normal, corner);
        w = n / dot(n, n);
        area = len(n);
    }
    bool is_interior(double a, double b) const {
        switch (shape) {
        case ORC_SHAPE_TRIANGLE: // :509-517
            return !(a < 0.0 || b < 0.0 || (a + b) > 1.0);
        case ORC_SHAPE_DISC: { // :627-638
            double ac = std::fabs(0.5 - a), bc = std::fabs(0.5 - b);
            return !(a < 0.0 || b < 0.0 || std::sqrt(ac * ac + bc * bc) > 0.5);
        }
        default: // :327-337
            return (0.0 <= a && a <= 1.0) && (0.0 <= b && b <= 1.0);
        }
    }
    std::optional<HitRecord> hit(const Ray &ray, Interval ray_t, Ctx &c) const override { // :341-371
        if (c.exclude_prim == prim) return std::nullopt;
        c.cnt[4]++;
        double denom = dot(normal, ray.d);
        if (std::fabs(denom) < 1e-8) return std::nullopt;
        double t = (d - dot(normal, ray.o)) / denom;
        if (!ray_t.contains(t)) return std::nullopt;
        c.cnt[5]++;
        V3 p = ray.at(t);
        V3 ph = p - corner;
        double alpha = dot(w, cross(ph, v));
        double beta = dot(w, cross(u, ph));
        if (!is_interior(alpha, beta)) return std::nullopt;
        return HitRecord::from_incoming_ray(ray, p, normal, t, alpha, beta, mat, prim);
    }
    const BBox *bounding_box() const override { return &bbox; }
    double pdf_value(V3 origin, V3 direction, Ctx &c) const override { // :377-389
        int save = c.exclude_prim;
        c.exclude_prim = -1;
        auto h = hit(Ray{origin, direction, 0.0}, Interval{0.001, INF}, c);
        c.exclude_prim = save;
        if (!h) return 0.0;
        double dist2 = h->t * h->t * len_squared(direction);
        double cosine = std::fabs(dot(direction, h->normal) / len(direction));
        return dist2 / (cosine * area);
    }
    V3 random(V3 origin, Ctx &c) const override { // :391-395
        double a = c.rng.next();
        double b = c.rng.next();
        V3 p = corner + (a * u) + (b * v);
        return p - origin;
    }
};

struct Translate : Hittable { // hittable.rs:679-716
    HPtr object;
    V3 offset;
    BBox bbox;
    Translate(HPtr o, V3 off) : object(std::move(o)), offset(off) { bbox = object->bounding_box()->shifted(off); }
    std::optional<HitRecord> hit(const Ray &ray, Interval ray_t, Ctx &c) const override {
        Ray r{ray.o - offset, ray.d, ray.time};
        auto h = object->hit(r, ray_t, c);
        if (!h) return std::nullopt;
        h->point = h->point + offset;
        return h;
    }
    const BBox *bounding_box() const override { return &bbox; }
};

struct RotateY : Hittable { // hittable.rs:718-818
    HPtr object;
    double sin_t, cos_t;
    BBox bbox;
    RotateY(HPtr o, double angle) : object(std::move(o)) {
        sin_t = std::sin(angle);
        cos_t = std::cos(angle);
        const BBox &b = *object->bounding_box();
        V3 mn{INF, INF, INF}, mx{-INF, -INF, -INF};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    double x = i * b.x.end + (1 - i) * b.x.start;
                    double y = j * b.y.end + (1 - j) * b.y.start;
                    double z = k * b.z.end + (1 - k) * b.z.start;
                    double nx = cos_t * x + sin_t * z;
                    double nz = -sin_t * x + cos_t * z;
                    V3 t{nx, y, nz};
                    for (int a = 0; a < 3; ++a) {
                        mn.at(a) = std::fmin(mn[a], t[a]);
                        mx.at(a) = std::fmax(mx[a], t[a]);
                    }
                }
        bbox = BBox::bounded_by(mn, mx);
    }
    std::optional<HitRecord> hit(const Ray &ray, Interval ray_t, Ctx &c) const override {
        V3 o{(cos_t * ray.o.x) - (sin_t * ray.o.z), ray.o.y, (sin_t * ray.o.x) + (cos_t * ray.o.z)};
        V3 d{(cos_t * ray.d.x) - (sin_t * ray.d.z), ray.d.y, (sin_t * ray.d.x) + (cos_t * ray.d.z)};
        auto h = object->hit(Ray{o, d, ray.time}, ray_t, c);
        if (!h) return std::nullopt;
        V3 p{(cos_t * h->point.x) + (sin_t * h->point.z), h->point.y, (-sin_t * h->point.x) + (cos_t * h->point.z)};
        V3 n{(cos_t * h->normal.x) + (sin_t * h->normal.z), h->normal.y,
             (-sin_t * h->normal.x) + (cos_t * h->normal.z)};
        h->point = p;
        h->normal = n;
        return h;
    }
    const BBox *bounding_box() const override { return &bbox; }
};

struct ConstantMedium : Hittable { // hittable.rs:820-891
    HPtr boundary;
    double neg_inv_density;
    std::unique_ptr<Isotropic> phase;
    int prim;
    std::optional<HitRecord> hit(const Ray &ray, Interval ray_t, Ctx &c) const override {
        if (c.exclude_prim == prim) return std::nullopt;
        int save = c.exclude_prim;
        c.exclude_prim = -1; // boundary primitives are not world primitives
        auto rec1 = boundary->hit(ray, Interval::universe(), c);
        std::optional<HitRecord> rec2;
        if (rec1) rec2 = boundary->hit(ray, Interval{rec1->t + 0.0001, INF}, c);
        c.exclude_prim = save;
        if (!rec1 || !rec2) return std::nullopt;
        if (rec1->t < ray_t.start) rec1->t = ray_t.start;
        if (rec2->t > ray_t.end) rec2->t = ray_t.end;
        if (rec1->t >= rec2->t) return std::nullopt;
        if (rec1->t < 0.0) rec1->t = 0.0;
        double ray_len = len(ray.d);
        double dist_inside = (rec2->t - rec1->t) * ray_len;
        double hit_dist = neg_inv_density * std::log(c.rng.medium_draw(c.bounce, uint32_t(prim)));
        if (hit_dist > dist_inside) return std::nullopt;
        double t = rec1->t + hit_dist / ray_len;
        HitRecord h;
        h.t = t;
        h.point = ray.at(t);
        h.normal = {1, 0, 0};
        h.front_face = true;
        h.material = phase.get();
        h.u = std::numeric_limits<double>::quiet_NaN();
        h.v = std::numeric_limits<double>::quiet_NaN();
        h.prim = prim;
        return h;
    }
    const BBox *bounding_box() const override { return boundary->bounding_box(); }
};

struct BVHNode : Hittable { // boundingbox.rs:171-236
    HPtr left, right;
    BBox bbox;
    explicit BVHNode(std::vector<HPtr> objects) {
        bbox = BBox::empty();
        for (auto &o : objects) { // extending_opt(Some(bbox), object.bounding_box()) :65-71
            const BBox *ob = o->bounding_box();
            bbox = ob ? BBox::extending(bbox, *ob) : BBox::make(bbox.x, bbox.y, bbox.z);
        }
        int axis = bbox.longest_axis();
        if (objects.size() == 1) {
            left = right = objects[0];
        } else if (objects.size() == 2) {
            left = objects[0];
            right = objects[1];
        } else {
            // sort_unstable_by on the interval START (:208-215); order among equal keys is
            // unspecified in the reference
            std::sort(objects.begin(), objects.end(), [axis](const HPtr &a, const HPtr &b) {
                return a->bounding_box()->ax(axis).start < b->bounding_box()->ax(axis).start;
            });
            size_t mid = objects.size() / 2;
            std::vector<HPtr> l(objects.begin(), objects.begin() + mid), r(objects.begin() + mid, objects.end());
            left = std::make_shared<BVHNode>(std::move(l));
            right = std::make_shared<BVHNode>(std::move(r));
        }
    }
    std::optional<HitRecord> hit(const Ray &ray, Interval ray_t, Ctx &c) const override { // :219-231
        c.cnt[1]++;
        if (!bbox.hit(ray, ray_t)) return std::nullopt;
        auto hl = left->hit(ray, ray_t, c);
        auto hr = hl ? right->hit(ray, Interval{ray_t.start, hl->t}, c) : right->hit(ray, ray_t, c);
        return hr ? hr : hl;
    }
    const BBox *bounding_box() const override { return &bbox; }
    void stats(int depth, int &nodes, int &maxdepth) const {
        nodes++;
        maxdepth = std::max(maxdepth, depth);
        if (auto l = dynamic_cast<const BVHNode *>(left.get())) l->stats(depth + 1, nodes, maxdepth);
        if (right != left)
            if (auto r = dynamic_cast<const BVHNode *>(right.get())) r->stats(depth + 1, nodes, maxdepth);
    }
};

// ---------------------------------------------------------------- PDFs (pdf.rs)
// cosine PDF: pdf.rs:20-37; sphere PDF: pdf.rs:9-18; hittable PDF: pdf.rs:38-60 (direct calls)
double cosine_pdf_value(const ONB &uvw, V3 dir) {
    double cos_theta = dot(unit(dir), uvw.w);
    return std::fmax(0.0, cos_theta / PI);
}

// ---------------------------------------------------------------- Camera (camera.rs)
struct Camera {
    uint32_t image_width, image_height;
    V3 center, pixel_00, pxdelta_u, pxdelta_v, disk_u, disk_v;
    uint32_t samples_per_px, sqrt_spp, max_depth;
    double px_sample_scale, sqrt_spp_scale, defocus_angle;
    bool sky;
    bool aa_disc = false;
    Color bg;

    explicit Camera(const orc_camera &b) { // camera.rs:280-366
        image_width = b.image_width;
        image_height = b.image_height;
        double fwidth = double(image_width), fheight = double(image_height);
        double aspect_ratio = fwidth / fheight;
        double vfov = b.vfov_deg * (PI / 180.0);
        defocus_angle = b.defocus_angle_deg * (PI / 180.0);
        double focal_length = b.focal_length;
        center = {b.center[0], b.center[1], b.center[2]};
        V3 target{b.target[0], b.target[1], b.target[2]};
        V3 vup = unit(V3{b.vup[0], b.vup[1], b.vup[2]});
        double viewport_height = 2.0 * std::tan(vfov / 2.0) * focal_length;
        double viewport_width = viewport_height * aspect_ratio;
        V3 w = unit(center - target);
        V3 u = cross(vup, w); // only *labelled* normalised in the reference (vec.rs:299-306)
        V3 v = cross(w, u);
        V3 viewport_u = viewport_width * u;
        V3 viewport_v = viewport_height * (-v);
        pxdelta_u = viewport_u / double(image_width);
        pxdelta_v = viewport_v / double(image_height);
        V3 corner = center - (focal_length * w) - viewport_u / 2.0 - viewport_v / 2.0;
        pixel_00 = corner + (pxdelta_u + pxdelta_v) / 2.0;
        double defocus_radius = focal_length * std::tan(defocus_angle / 2.0);
        disk_u = u * defocus_radius;
        disk_v = v * defocus_radius;
        sqrt_spp = uint32_t(std::sqrt(double(b.samples_per_px)));
        samples_per_px = sqrt_spp * sqrt_spp;
        px_sample_scale = 1.0 / double(samples_per_px);
        sqrt_spp_scale = 1.0 / double(sqrt_spp);
        max_depth = b.max_depth;
        sky = b.background_sky != 0;
        aa_disc = b.antialiasing == 1;
        bg = {b.background[0], b.background[1], b.background[2]};
    }

    Ray get_ray(uint32_t i, uint32_t j, uint32_t si, uint32_t sj, Ctx &c) const { // camera.rs:409-430
        double ox, oy;
        if (aa_disc) { // documented at camera.rs:209-210, drawn like two_d.rs:183-190
            for (;;) {
                ox = c.rng.range_pm1();
                oy = c.rng.range_pm1();
                if (ox * ox + oy * oy < 1.0) break;
            }
            ox *= 0.5, oy *= 0.5;
        } else {
            ox = ((double(si) + c.rng.next()) * sqrt_spp_scale) - 0.5;
            oy = ((double(sj) + c.rng.next()) * sqrt_spp_scale) - 0.5;
        }
        V3 px_sample = pixel_00 + (double(i) + ox) * pxdelta_u + (double(j) + oy) * pxdelta_v;
        V3 origin = center;
        if (!(defocus_angle <= 0.0)) { // camera.rs:484-488, two_d.rs:183-190
            double px, py;
            for (;;) {
                px = c.rng.range_pm1();
                py = c.rng.range_pm1();
                if (px * px + py * py < 1.0) break;
            }
            origin = center + px * disk_u + py * disk_v;
        }
        V3 dir = px_sample - origin;
        double time = c.rng.next();
        return Ray{origin, dir, time};
    }

    static Color skybox(const Ray &ray) { // camera.rs:473-482
        V3 nd = unit(ray.d);
        double intensity = (nd.y + 1.0) * 0.5;
        V3 white = V3{1, 1, 1} * (1.0 - intensity);
        V3 col = V3{0.5, 0.7, 1.0} * intensity;
        V3 s = white + col;
        return {s.x, s.y, s.z};
    }
};

struct Renderer {
    const Camera &cam;
    const Hittable *world;
    const Hittable *lights; // may be null
    Ctx &c;

    Color ray_color(const Ray &ray, uint32_t depth, bool dead = false) { // camera.rs:432-471
        if (depth == 0) return {};
        c.bounce = cam.max_depth - depth;
        c.cnt[0]++;
        if (dead) c.cnt[8]++;
        // Intersection-test counters (cnt[1..5]) cover the segments that carry weight: the ones the device
        // traces too, so work per segment is averaged over one population (segments - zero-weight segments).
        uint64_t keep[5];
        if (dead) std::copy(c.cnt + 1, c.cnt + 6, keep);
        auto hit = world->hit(ray, Interval{0.001, INF}, c);
        if (dead) std::copy(keep, keep + 5, c.cnt + 1);
        if (!hit) return cam.sky ? Camera::skybox(ray) : cam.bg;
        const Material *m = hit->material;
        Color emission = m->emitted(ray, *hit, hit->u, hit->v, hit->point);
        c.rng.begin(STREAM_SHADE, c.bounce);
        auto sc = m->scatter(ray, *hit, c);
        if (!sc || c.error) return emission;
        if (hit->sphere_radius > 0 && hit->sphere_radius < c.flag_radius) c.path_flag = true; // orc_render_flagged
        if (c.mode == ORC_MODE_BOOK3) return book3_tail(ray, *hit, *sc, emission, depth, dead);

        // shipped estimator: light-only sampling, BSDF pdf on the material's own sample
        if (!lights) {
            c.error = 2;
            return emission;
        }
        V3 dir = lights->random(hit->point, c);                // pdf.rs:57-59
        Ray scattered{hit->point, dir, ray.time};              // camera.rs:461
        double pdf_value = lights->pdf_value(hit->point, dir, c); // camera.rs:462
        double spdf = m->scattering_pdf(ray, *hit, sc->scattered, c); // camera.rs:464
        double weight = spdf / pdf_value;
        if (c.sanitize && !(pdf_value > 0.0 && std::isfinite(weight))) {
            c.cnt[6]++;
            return emission;
        }
        Color sample = ray_color(scattered, depth - 1, dead || weight == 0.0);
        Color scatter_color = cscale(cmul(sc->attenuation, sample), weight); // :467-468
        return cadd(emission, scatter_color);
    }

    // Mixture estimator of "The Rest of Your Life" ch.12, assembled from the reference's own
    // pdf.rs / onb.rs / material.rs parts.  Not wired in the reference: PARITY UNPINNED.
    Color book3_tail(const Ray &ray, const HitRecord &hit, const MaterialResult &sc, Color emission, uint32_t depth,
                     bool dead) {
        if (sc.skip_pdf) return cmul(sc.attenuation, ray_color(sc.scattered, depth - 1, dead));
        ONB uvw(hit.normal);
        auto mat_generate = [&]() -> V3 {
            if (sc.pdf_kind == PDF_SPHERE) return random_in_unit_sphere(c); // pdf.rs:15-17
            return uvw.transform(random_on_sphere_cosine(c));               // pdf.rs:34-36
        };
        auto mat_value = [&](V3 d) -> double {
            if (sc.pdf_kind == PDF_SPHERE) return 1.0 / (4.0 * PI); // pdf.rs:11-13
            return cosine_pdf_value(uvw, d);                        // pdf.rs:29-32
        };
        V3 dir;
        double pdf_value;
        if (lights) {
            // MixedPDF::generate is inverted in the reference (pdf.rs:92-102); intended form
            double xi = c.rng.next();
            dir = xi < 0.5 ? lights->random(hit.point, c) : mat_generate();
            pdf_value = 0.5 * lights->pdf_value(hit.point, dir, c) + 0.5 * mat_value(dir); // pdf.rs:85-90
        } else {
            dir = mat_generate();
            pdf_value = mat_value(dir);
        }
        Ray scattered{hit.point, dir, ray.time};
        double spdf = hit.material->scattering_pdf(ray, hit, scattered, c);
        double weight = spdf / pdf_value;
        if (c.sanitize && !(pdf_value > 0.0 && std::isfinite(weight))) {
            c.cnt[6]++;
            return emission;
        }
        Color sample = ray_color(scattered, depth - 1, dead || weight == 0.0);
        return cadd(emission, cscale(cmul(sc.attenuation, sample), weight));
    }
};

} // namespace

// ================================================================= C interface
struct orc_scene {
    std::vector<std::shared_ptr<Texture>> textures;
    std::vector<std::shared_ptr<Material>> materials;
    std::vector<HPtr> objects;
    HPtr world, lights;
    int nprims = 0;
    std::string err;
};

namespace {
V3 v3(const double *p) { return {p[0], p[1], p[2]}; }
bool ok_tex(orc_scene *s, int t) { return t >= 0 && t < int(s->textures.size()); }
bool ok_mat(orc_scene *s, int m) { return m >= 0 && m < int(s->materials.size()); }
bool ok_obj(orc_scene *s, int o) { return o >= 0 && o < int(s->objects.size()); }
int fail(orc_scene *s, const char *msg) {
    s->err = msg;
    return -1;
}
int push_obj(orc_scene *s, HPtr o) {
    s->objects.push_back(std::move(o));
    return int(s->objects.size()) - 1;
}
void init_ctx(Ctx &c, const orc_render_params &p) {
    c.mode = p.mode;
    c.sanitize = p.sanitize != 0;
    c.rng.kind = p.rng_kind;
    c.rng.key[0] = uint32_t(p.seed);
    c.rng.key[1] = uint32_t(p.seed >> 32);
}
} // namespace

extern "C" {

orc_scene *orc_scene_new(void) { return new orc_scene(); }
void orc_scene_free(orc_scene *s) { delete s; }
const char *orc_last_error(orc_scene *s) { return s->err.c_str(); }

int orc_tex_solid(orc_scene *s, double r, double g, double b) {
    s->textures.push_back(std::make_shared<SolidColor>(Color{r, g, b}));
    return int(s->textures.size()) - 1;
}
int orc_tex_checker(orc_scene *s, double scale, int even, int odd) {
    if (!ok_tex(s, even) || !ok_tex(s, odd)) return fail(s, "checker: bad texture id");
    auto t = std::make_shared<Checkerboard>();
    t->scale = scale;
    t->even = s->textures[even];
    t->odd = s->textures[odd];
    s->textures.push_back(t);
    return int(s->textures.size()) - 1;
}
int orc_tex_image(orc_scene *s, uint32_t w, uint32_t h, const uint8_t *rgb) {
    if (w == 0 || h == 0) return fail(s, "image: empty");
    auto t = std::make_shared<ImageTexture>();
    t->w = w;
    t->h = h;
    t->data.assign(rgb, rgb + size_t(w) * h * 3);
    s->textures.push_back(t);
    return int(s->textures.size()) - 1;
}
int orc_tex_noise(orc_scene *s, double scale, const double *ranvec, const int32_t *perm) {
    auto t = std::make_shared<NoiseTexture>();
    t->scale = scale;
    for (int i = 0; i < 256; ++i) {
        t->ranvec[i] = {ranvec[3 * i], ranvec[3 * i + 1], ranvec[3 * i + 2]};
        t->px[i] = perm[i];
        t->py[i] = perm[256 + i];
        t->pz[i] = perm[512 + i];
    }
    s->textures.push_back(t);
    return int(s->textures.size()) - 1;
}

int orc_mat_lambertian(orc_scene *s, int tex) {
    if (!ok_tex(s, tex)) return fail(s, "lambertian: bad texture id");
    auto m = std::make_shared<Lambertian>();
    m->tex = s->textures[tex];
    s->materials.push_back(m);
    return int(s->materials.size()) - 1;
}
int orc_mat_light(orc_scene *s, int tex) {
    if (!ok_tex(s, tex)) return fail(s, "light: bad texture id");
    auto m = std::make_shared<DiffuseLight>();
    m->tex = s->textures[tex];
    s->materials.push_back(m);
    return int(s->materials.size()) - 1;
}
int orc_mat_metal(orc_scene *s, double r, double g, double b, double fuzz) {
    if (!(fuzz >= 0.0 && fuzz <= 1.0)) return fail(s, "Invalid fuzz value (expected 0.0..=1.0)"); // material.rs:106-109
    auto m = std::make_shared<Metal>();
    m->albedo = {r, g, b};
    m->fuzz = fuzz;
    s->materials.push_back(m);
    return int(s->materials.size()) - 1;
}
int orc_mat_dielectric(orc_scene *s, double ior) {
    auto m = std::make_shared<Dielectric>();
    m->ir = ior;
    s->materials.push_back(m);
    return int(s->materials.size()) - 1;
}
int orc_mat_isotropic(orc_scene *s, int tex) {
    if (!ok_tex(s, tex)) return fail(s, "isotropic: bad texture id");
    auto m = std::make_shared<Isotropic>();
    m->tex = s->textures[tex];
    s->materials.push_back(m);
    return int(s->materials.size()) - 1;
}

int orc_sphere(orc_scene *s, const double c0[3], const double cv[3], double radius, int mat) {
    if (!ok_mat(s, mat)) return fail(s, "sphere: bad material id");
    if (!(radius >= 0.0)) return fail(s, "sphere: radius must be >= 0"); // hittable.rs:131
    return push_obj(s, std::make_shared<Sphere>(v3(c0), v3(cv), radius, s->materials[mat].get(), s->nprims++));
}
int orc_quad(orc_scene *s, int shape, const double q[3], const double u[3], const double v[3], int mat) {
    if (!ok_mat(s, mat)) return fail(s, "quad: bad material id");
    if (shape < 0 || shape > 2) return fail(s, "quad: bad shape");
    return push_obj(s, std::make_shared<Quad>(shape, v3(q), v3(u), v3(v), s->materials[mat].get(), s->nprims++));
}
int orc_box3(orc_scene *s, const double a[3], const double b[3], int mat) { // hittable.rs:398-455
    if (!ok_mat(s, mat)) return fail(s, "box3: bad material id");
    const Material *m = s->materials[mat].get();
    V3 mn{std::fmin(a[0], b[0]), std::fmin(a[1], b[1]), std::fmin(a[2], b[2])};
    V3 mx{std::fmax(a[0], b[0]), std::fmax(a[1], b[1]), std::fmax(a[2], b[2])};
    V3 dx{mx.x - mn.x, 0, 0}, dy{0, mx.y - mn.y, 0}, dz{0, 0, mx.z - mn.z};
    auto sides = std::make_shared<HittableVec>();
    auto q = [&](V3 c, V3 u, V3 v) {
        sides->add(std::make_shared<Quad>(ORC_SHAPE_PARALLELOGRAM, c, u, v, m, s->nprims++));
    };
    q({mn.x, mn.y, mx.z}, dx, dy);
    q({mx.x, mn.y, mx.z}, -dz, dy);
    q({mx.x, mn.y, mn.z}, -dx, dy);
    q({mn.x, mn.y, mn.z}, dz, dy);
    q({mn.x, mx.y, mx.z}, dx, -dz);
    q({mn.x, mn.y, mn.z}, dx, dz);
    return push_obj(s, sides);
}
int orc_list(orc_scene *s, const int *objs, int n) {
    auto l = std::make_shared<HittableVec>();
    for (int i = 0; i < n; ++i) {
        if (!ok_obj(s, objs[i])) return fail(s, "list: bad object id");
        if (!s->objects[objs[i]]->bounding_box()) return fail(s, "list: object without bounding box");
        l->add(s->objects[objs[i]]);
    }
    return push_obj(s, l);
}
int orc_bvh(orc_scene *s, const int *objs, int n) {
    if (n <= 0) return fail(s, "bvh: empty object list"); // the reference recurses forever here
    std::vector<HPtr> v;
    for (int i = 0; i < n; ++i) {
        if (!ok_obj(s, objs[i])) return fail(s, "bvh: bad object id");
        if (!s->objects[objs[i]]->bounding_box()) return fail(s, "bvh: object without bounding box");
        v.push_back(s->objects[objs[i]]);
    }
    return push_obj(s, std::make_shared<BVHNode>(std::move(v)));
}
int orc_translate(orc_scene *s, int obj, const double off[3]) {
    if (!ok_obj(s, obj)) return fail(s, "translate: bad object id");
    if (!s->objects[obj]->bounding_box()) return fail(s, "Objects without bounding boxes should not be Translated");
    return push_obj(s, std::make_shared<Translate>(s->objects[obj], v3(off)));
}
int orc_rotate_y(orc_scene *s, int obj, double angle) {
    if (!ok_obj(s, obj)) return fail(s, "rotate_y: bad object id");
    if (!s->objects[obj]->bounding_box()) return fail(s, "Objects without bounding boxes should not be Rotated");
    return push_obj(s, std::make_shared<RotateY>(s->objects[obj], angle));
}
int orc_medium(orc_scene *s, int boundary, double density, int tex) { // hittable.rs:828-835
    if (!ok_obj(s, boundary)) return fail(s, "medium: bad object id");
    if (!ok_tex(s, tex)) return fail(s, "medium: bad texture id");
    auto m = std::make_shared<ConstantMedium>();
    m->boundary = s->objects[boundary];
    m->neg_inv_density = -1.0 / density;
    m->phase = std::make_unique<Isotropic>();
    m->phase->tex = s->textures[tex];
    m->prim = s->nprims++;
    return push_obj(s, m);
}
int orc_set_world(orc_scene *s, int obj) {
    if (!ok_obj(s, obj)) return fail(s, "set_world: bad object id");
    s->world = s->objects[obj];
    return 0;
}
int orc_set_lights(orc_scene *s, int obj) {
    if (obj < 0) {
        s->lights.reset();
        return 0;
    }
    if (!ok_obj(s, obj)) return fail(s, "set_lights: bad object id");
    s->lights = s->objects[obj];
    return 0;
}
int orc_prim_count(orc_scene *s) { return s->nprims; }

namespace {
int render_impl(orc_scene *s, const orc_camera *cb, const orc_render_params *p, double *film_sum, double *film_sumsq,
                uint64_t *counters, double flag_radius, uint32_t *flagged);
}
int orc_render(orc_scene *s, const orc_camera *cb, const orc_render_params *p, double *film_sum,
               double *film_sumsq, uint64_t *counters) {
    return render_impl(s, cb, p, film_sum, film_sumsq, counters, 0.0, nullptr);
}
int orc_render_flagged(orc_scene *s, const orc_camera *cb, const orc_render_params *p, double *film_sum, double radius_below,
                       uint32_t *flagged) {
    if (!flagged) return fail(s, "render_flagged: null output");
    Camera cam(*cb);
    std::memset(flagged, 0, sizeof(uint32_t) * size_t(cam.image_width) * cam.image_height);
    return render_impl(s, cb, p, film_sum, nullptr, nullptr, radius_below, flagged);
}
namespace {
int render_impl(orc_scene *s, const orc_camera *cb, const orc_render_params *p, double *film_sum, double *film_sumsq,
                uint64_t *counters, double flag_radius, uint32_t *flagged) {
    if (!s->world) return fail(s, "render: no world set");
    Camera cam(*cb);
    if (cam.sqrt_spp == 0) return fail(s, "render: samples_per_px < 1");
    uint32_t W = cam.image_width, H = cam.image_height;
    uint32_t sb = p->stratum_begin, se = p->stratum_end ? p->stratum_end : cam.samples_per_px;
    if (se > cam.samples_per_px || sb > se) return fail(s, "render: bad stratum range");
    uint32_t x0 = p->x0, y0 = p->y0, x1 = p->x1 ? p->x1 : W, y1 = p->x1 ? p->y1 : H;
    if (x1 > W || y1 > H) return fail(s, "render: bad window");
    std::memset(film_sum, 0, sizeof(double) * size_t(W) * H * 3);
    if (film_sumsq) std::memset(film_sumsq, 0, sizeof(double) * size_t(W) * H * 3);
    uint64_t total[ORC_NCOUNTERS] = {0};
    int error = 0;
    int threads = p->threads > 1 ? p->threads : 1;

#pragma omp parallel num_threads(threads) if (threads > 1)
    {
        Ctx c;
        init_ctx(c, *p);
        c.flag_radius = flag_radius;
        if (c.rng.kind == ORC_RNG_MT64) c.rng.mt.seed(p->seed);
        Renderer R{cam, s->world.get(), s->lights.get(), c};
#pragma omp for schedule(dynamic, 1)
        for (int64_t j = y0; j < int64_t(y1); ++j) { // camera.rs:386-401
            if (c.rng.kind == ORC_RNG_MT64 && threads > 1) c.rng.mt.seed(p->seed * 1000003ull + uint64_t(j));
            for (uint32_t i = x0; i < x1; ++i) {
                size_t px = (size_t(j) * W + i) * 3;
                for (uint32_t sidx = sb; sidx < se; ++sidx) {
                    uint32_t sj = sidx / cam.sqrt_spp, si = sidx % cam.sqrt_spp;
                    c.rng.pixel = uint32_t(j) * W + i;
                    c.rng.sample = sidx;
                    c.rng.begin(STREAM_CAMERA, 0);
                    Ray ray = cam.get_ray(i, uint32_t(j), si, sj, c);
                    c.cnt[7]++;
                    c.path_flag = false;
                    Color col = R.ray_color(ray, cam.max_depth);
                    if (flagged && c.path_flag) flagged[size_t(j) * W + i] += 1; // rows belong to one thread each
                    film_sum[px] += col.r;
                    film_sum[px + 1] += col.g;
                    film_sum[px + 2] += col.b;
                    if (film_sumsq) {
                        film_sumsq[px] += col.r * col.r;
                        film_sumsq[px + 1] += col.g * col.g;
                        film_sumsq[px + 2] += col.b * col.b;
                    }
                }
            }
        }
#pragma omp critical
        {
            for (int k = 0; k < ORC_NCOUNTERS; ++k) total[k] += c.cnt[k];
            if (c.error) error = c.error;
        }
    }
    if (counters) std::memcpy(counters, total, sizeof(total));
    if (error == 1) return fail(s, "render: material cannot scatter in this mode (reference: todo!()/unimplemented!())");
    if (error == 2) return fail(s, "render: lights object lacks pdf_value/random (reference: unimplemented!())");
    return 0;
}
} // namespace

int orc_camera_rays(orc_scene *s, const orc_camera *cb, const orc_render_params *p, uint32_t n,
                    const uint32_t *pixel, const uint32_t *stratum, double *o, double *d, double *time) {
    Camera cam(*cb);
    if (cam.sqrt_spp == 0) return fail(s, "camera_rays: samples_per_px < 1");
    Ctx c;
    init_ctx(c, *p);
    if (c.rng.kind == ORC_RNG_MT64) c.rng.mt.seed(p->seed);
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t i = pixel[k] % cam.image_width, j = pixel[k] / cam.image_width;
        uint32_t sj = stratum[k] / cam.sqrt_spp, si = stratum[k] % cam.sqrt_spp;
        c.rng.pixel = pixel[k];
        c.rng.sample = stratum[k];
        c.rng.begin(STREAM_CAMERA, 0);
        Ray r = cam.get_ray(i, j, si, sj, c);
        o[3 * k] = r.o.x, o[3 * k + 1] = r.o.y, o[3 * k + 2] = r.o.z;
        d[3 * k] = r.d.x, d[3 * k + 1] = r.d.y, d[3 * k + 2] = r.d.z;
        time[k] = r.time;
    }
    return 0;
}

int orc_intersect_batch(orc_scene *s, uint32_t n, const double *o, const double *d, const double *time,
                        double tmin, double tmax, int32_t *hit, int32_t *prim, double *t, double *u,
                        double *v, double *second_t, double *normal) {
    if (!s->world) return fail(s, "intersect_batch: no world set");
    Ctx c;
    c.rng.kind = ORC_RNG_PHILOX24;
    for (uint32_t k = 0; k < n; ++k) {
        Ray r{v3(o + 3 * k), v3(d + 3 * k), time ? time[k] : 0.0};
        c.rng.pixel = k; // medium draws are keyed by ray index in batch mode
        c.rng.sample = 0;
        c.bounce = 0;
        c.exclude_prim = -1;
        auto h = s->world->hit(r, Interval{tmin, tmax}, c);
        hit[k] = h ? 1 : 0;
        prim[k] = h ? h->prim : -1;
        t[k] = h ? h->t : INF;
        if (u) u[k] = h ? h->u : 0.0;
        if (v) v[k] = h ? h->v : 0.0;
        if (normal) {
            V3 nn = h ? h->normal : V3{0, 0, 0};
            normal[3 * k] = nn.x, normal[3 * k + 1] = nn.y, normal[3 * k + 2] = nn.z;
        }
        if (second_t) {
            second_t[k] = INF;
            if (h) {
                c.exclude_prim = h->prim;
                auto h2 = s->world->hit(r, Interval{tmin, tmax}, c);
                if (h2) second_t[k] = h2->t;
            }
        }
    }
    return 0;
}

void orc_sphere_uv(const double p[3], double uv[2]) { sphere_get_uv(v3(p), uv[0], uv[1]); }
void orc_onb(const double n[3], double out[9]) {
    ONB b(v3(n));
    const V3 *a[3] = {&b.u, &b.v, &b.w};
    for (int i = 0; i < 3; ++i) out[3 * i] = a[i]->x, out[3 * i + 1] = a[i]->y, out[3 * i + 2] = a[i]->z;
}
void orc_cosine_sample(double r1, double r2, double out[3]) {
    V3 p = cosine_from(r1, r2);
    out[0] = p.x, out[1] = p.y, out[2] = p.z;
}
void orc_reflect(const double v[3], const double n[3], double out[3]) {
    V3 r = reflect(v3(v), v3(n));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_refract(const double uv[3], const double n[3], double ratio, double out[3]) {
    V3 r = refract(v3(uv), v3(n), ratio);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
double orc_schlick(double cosine, double ri) { return schlick(cosine, ri); }
void orc_color_hex(uint32_t hex, double rgb[3]) {
    Color c = color_hex(hex);
    rgb[0] = c.r, rgb[1] = c.g, rgb[2] = c.b;
}
void orc_color_quantize(const double rgb[3], int gamma, uint8_t out[3]) { // color.rs:130-155, 209-215
    for (int i = 0; i < 3; ++i) {
        double x = rgb[i];
        if (gamma) x = x > 0.0 ? std::sqrt(x) : 0.0;
        // Interval::clamp: NaN fails both comparisons and passes through; `as u8` maps NaN to 0
        double cl = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
        double s = cl * 255.0;
        out[i] = (s != s) ? 0 : uint8_t(s);
    }
}
int orc_interval_contains(double a, double b, double x) { return Interval{a, b}.contains(x); }
int orc_interval_surrounds(double a, double b, double x) { return Interval{a, b}.surrounds(x); }
int orc_interval_overlap(double a0, double a1, double b0, double b1, double out[2]) {
    auto r = Interval{a0, a1}.overlap(Interval{b0, b1});
    if (!r) return 0;
    out[0] = r->start, out[1] = r->end;
    return 1;
}
void orc_interval_expand(double a, double b, double delta, double out[2]) {
    Interval r = Interval{a, b}.expand(delta);
    out[0] = r.start, out[1] = r.end;
}
double orc_interval_clamp(double a, double b, double x) { return Interval{a, b}.clamp(x); }
void orc_tex_value(orc_scene *s, int tex, double u, double v, const double p[3], double rgb[3]) {
    Color c = s->textures[tex]->value(u, v, v3(p));
    rgb[0] = c.r, rgb[1] = c.g, rgb[2] = c.b;
}
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
void orc_philox2x32(const uint32_t ctr[2], uint32_t key, uint32_t out[2]) { Sampler::philox2x32_10(ctr[0], ctr[1], key, out); }
int orc_aabb_hit(const double bmin[3], const double bmax[3], const double o[3], const double d[3],
                 double tmin, double tmax) {
    BBox b{{bmin[0], bmax[0]}, {bmin[1], bmax[1]}, {bmin[2], bmax[2]}};
    return b.hit(Ray{v3(o), v3(d), 0.0}, Interval{tmin, tmax}) ? 1 : 0;
}
int orc_bvh_stats(orc_scene *s, int obj, int *nodes, int *depth) {
    if (!ok_obj(s, obj)) return fail(s, "bvh_stats: bad object id");
    auto b = dynamic_cast<const BVHNode *>(s->objects[obj].get());
    if (!b) return fail(s, "bvh_stats: not a BVH");
    *nodes = 0;
    *depth = 0;
    b->stats(1, *nodes, *depth);
    return 0;
}

} // extern "C"
