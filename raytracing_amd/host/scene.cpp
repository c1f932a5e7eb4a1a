// scene.cpp -- host scene graph: math, bounding boxes, primitives, BVH build, camera set-up.
#include "raytracing.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <random>

namespace raytracing {

namespace {
constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double INF = std::numeric_limits<double>::infinity();
} // namespace

// ------------------------------------------------------------------ math
double Vec3::len() const { return std::sqrt(len_squared()); }
Vec3 Vec3::as_unit() const { return *this / len(); }
double Vec3::dot(const Vec3 &a, const Vec3 &b) { return std::fma(a.x, b.x, std::fma(a.y, b.y, a.z * b.z)); }
Vec3 Vec3::cross(const Vec3 &a, const Vec3 &b) {
    return {std::fma(a.y, b.z, -(a.z * b.y)), std::fma(a.z, b.x, -(a.x * b.z)), std::fma(a.x, b.y, -(a.y * b.x))};
}

Color Color::hex(uint32_t color) {
    uint32_t r, g, b;
    if (color <= 0xfff) {
        r = ((color & 0xf00) >> 8) * 16;
        g = (((color << 4) & 0xf00) >> 8) * 16;
        b = (((color << 8) & 0xf00) >> 8) * 16;
    } else {
        r = (color & 0xff0000) >> 16;
        g = ((color << 8) & 0xff0000) >> 16;
        b = ((color << 16) & 0xff0000) >> 16;
    }
    return {double(r) / 255.0, double(g) / 255.0, double(b) / 255.0};
}
bool Color::is_valid() const {
    Interval i{0.0, 1.0};
    return i.contains(r) && i.contains(g) && i.contains(b);
}
std::array<uint8_t, 3> Color::as_rgb_ints() const {
    auto q = [](double x) -> uint8_t {
        double c = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); // NaN passes through Interval::clamp
        double s = c * 255.0;
        return s != s ? uint8_t(0) : uint8_t(s); // `as u8`: NaN -> 0, truncation otherwise
    };
    return {q(r), q(g), q(b)};
}
Color Color::as_gamma_corrected() const {
    auto g2 = [](double x) { return x > 0.0 ? std::sqrt(x) : 0.0; };
    return {g2(r), g2(g), g2(b)};
}

Interval Interval::empty() { return {INF, -INF}; }
Interval Interval::universe() { return {-INF, INF}; }
Interval Interval::surrounding(const Interval &a, const Interval &b) {
    return positive(std::fmin(a.start, b.start), std::fmax(a.end, b.end));
}
std::optional<Interval> Interval::overlap(const Interval &o) const {
    Interval r{std::fmax(start, o.start), std::fmin(end, o.end)};
    if (r.size() <= 0.0) return std::nullopt;
    return r;
}

BoundingBox3::BoundingBox3(Interval x_, Interval y_, Interval z_) {
    const double DELTA = 0.0001;
    auto expand = [&](Interval i) { return i.size() < DELTA ? i.expand(DELTA) : i; };
    x = expand(x_), y = expand(y_), z = expand(z_);
}
BoundingBox3 BoundingBox3::bounded_by(const Point3 &a, const Point3 &b) {
    return BoundingBox3(Interval::positive(a.x, b.x), Interval::positive(a.y, b.y), Interval::positive(a.z, b.z));
}
BoundingBox3 BoundingBox3::extending(const BoundingBox3 &a, const BoundingBox3 &b) {
    return BoundingBox3(Interval::surrounding(a.x, b.x), Interval::surrounding(a.y, b.y), Interval::surrounding(a.z, b.z));
}
int BoundingBox3::longest_axis() const {
    int ax = 0;
    double best = 0.0;
    for (int i = 0; i < 3; ++i)
        if (axis(i).size() > best) best = axis(i).size(), ax = i;
    return ax;
}
BoundingBox3 BoundingBox3::operator+(const Vec3 &o) const {
    return BoundingBox3({x.start + o.x, x.end + o.x}, {y.start + o.y, y.end + o.y}, {z.start + o.z, z.end + o.z});
}

// ------------------------------------------------------------------ textures / materials
std::shared_ptr<Checkerboard> Checkerboard::solid(double s, Color even, Color odd) {
    return std::make_shared<Checkerboard>(s, std::make_shared<SolidColor>(even), std::make_shared<SolidColor>(odd));
}
ImageTexture::ImageTexture(std::vector<uint8_t> data, uint32_t w, uint32_t h)
    : image_data(std::move(data)), width(w), height(h) {
    if (w == 0 || h == 0 || image_data.size() != size_t(w) * h * 3)
        throw Error("sanity check; image_data.len() = # of pixels * 3 channels per pixel"); // texture.rs:83-87
}
std::shared_ptr<ImageTexture> ImageTexture::load(const std::string &path) {
    std::vector<uint8_t> rgb;
    uint32_t w, h;
    read_png_rgb8(path, rgb, w, h);
    return std::make_shared<ImageTexture>(std::move(rgb), w, h);
}
NoiseTexture::NoiseTexture(double s, uint64_t seed) : scale(s) {
    std::mt19937_64 rng(seed);
    auto uni = [&]() { return double(rng() >> 11) * (1.0 / 9007199254740992.0); };
    for (auto &v : ranvec) v = Vec3(2 * uni() - 1, 2 * uni() - 1, 2 * uni() - 1).as_unit();
    for (auto &p : perm) {
        for (int i = 0; i < 256; ++i) p[i] = i;
        for (int i = 255; i > 0; --i) std::swap(p[i], p[int(uni() * (i + 1))]);
    }
}
NoiseTexture::NoiseTexture(double s, const double *rv, const int32_t *pm) : scale(s) {
    for (int i = 0; i < 256; ++i) ranvec[i] = {rv[3 * i], rv[3 * i + 1], rv[3 * i + 2]};
    for (int a = 0; a < 3; ++a)
        for (int i = 0; i < 256; ++i) {
            if (pm[a * 256 + i] < 0 || pm[a * 256 + i] > 255) throw Error("noise permutation entry outside 0..255");
            perm[a][i] = pm[a * 256 + i];
        }
}

std::shared_ptr<Lambertian> Lambertian::solid(Color c) { return std::make_shared<Lambertian>(std::make_shared<SolidColor>(c)); }
std::shared_ptr<DiffuseLight> DiffuseLight::solid(Color c) { return std::make_shared<DiffuseLight>(std::make_shared<SolidColor>(c)); }
std::shared_ptr<Metal> Metal::with_fuzz(Color albedo, double fuzz) {
    if (!(fuzz >= 0.0 && fuzz <= 1.0)) throw Error("Invalid fuzz value (expected 0.0..=1.0)"); // material.rs:106-109
    auto m = std::make_shared<Metal>(albedo);
    m->fuzz = fuzz;
    return m;
}
std::shared_ptr<Isotropic> Isotropic::colored(Color c) { return std::make_shared<Isotropic>(std::make_shared<SolidColor>(c)); }

// ------------------------------------------------------------------ hittables
Sphere::Sphere(Point3 c0, Vec3 cv, double r, MaterialPtr m) : center0(c0), center_vec(cv), radius(r), material(std::move(m)) {
    if (!(r >= 0.0)) throw Error("assertion failed: radius >= 0.0"); // hittable.rs:131
    Vec3 rad{r, r, r};
    BoundingBox3 b0 = BoundingBox3::bounded_by(c0 - rad, c0 + rad);
    Point3 c1 = c0 + cv * 1.0;
    BoundingBox3 b1 = BoundingBox3::bounded_by(c1 - rad, c1 + rad);
    bbox = BoundingBox3::extending(b0, b1);
}
std::shared_ptr<Sphere> Sphere::stationary(Point3 c, double r, MaterialPtr m) {
    return std::make_shared<Sphere>(c, Vec3{}, r, std::move(m));
}

Parallelogram::Parallelogram(Point3 q, Vec3 u_, Vec3 v_, MaterialPtr m, uint32_t shape_)
    : shape(shape_), corner(q), u(u_), v(v_), material(std::move(m)) {
    BoundingBox3 d1 = BoundingBox3::bounded_by(corner, corner + u + v); // hittable.rs:296-299
    BoundingBox3 d2 = BoundingBox3::bounded_by(corner + u, corner + v);
    bbox = BoundingBox3::extending(d1, d2);
}
std::shared_ptr<Parallelogram> Triangle(Point3 c, Vec3 u, Vec3 v, MaterialPtr m) {
    return std::make_shared<Parallelogram>(c, u, v, std::move(m), RT_SHAPE_TRIANGLE);
}
std::shared_ptr<Parallelogram> TriangleFromPoints(Point3 a, Point3 b, Point3 c, MaterialPtr m) { // hittable.rs:494-501
    return Triangle(a, b - a, c - a, std::move(m));
}
std::shared_ptr<Parallelogram> Disc(Point3 c, Vec3 u, Vec3 v, MaterialPtr m) {
    return std::make_shared<Parallelogram>(c, u, v, std::move(m), RT_SHAPE_DISC);
}
std::shared_ptr<Parallelogram> DiscFromCenter(Point3 center, Vec3 u, Vec3 v, MaterialPtr m) {
    // hittable.rs:596-619: corner = center - u - v, edges 2u and 2v (same box as the reference's)
    return Disc(center - u - v, u * 2.0, v * 2.0, std::move(m));
}

void HittableVec::add(HittablePtr obj) {
    const BoundingBox3 *ob = obj ? obj->bounding_box() : nullptr;
    if (!ob) throw Error("called `Option::unwrap()` on a `None` value: object has no bounding box"); // hittable.rs:240
    bbox = bbox ? BoundingBox3::extending(*bbox, *ob) : *ob;
    objects.push_back(std::move(obj));
}

std::shared_ptr<HittableVec> box3(const Point3 &a, const Point3 &b, MaterialPtr mat) {
    auto sides = std::make_shared<HittableVec>();
    Point3 mn{std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)};
    Point3 mx{std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)};
    Vec3 dx{mx.x - mn.x, 0, 0}, dy{0, mx.y - mn.y, 0}, dz{0, 0, mx.z - mn.z};
    auto q = [&](Point3 c, Vec3 u, Vec3 v) { sides->add(std::make_shared<Parallelogram>(c, u, v, mat)); };
    q({mn.x, mn.y, mx.z}, dx, dy);  // front
    q({mx.x, mn.y, mx.z}, -dz, dy); // right
    q({mx.x, mn.y, mn.z}, -dx, dy); // back
    q({mn.x, mn.y, mn.z}, dz, dy);  // left
    q({mn.x, mx.y, mx.z}, dx, -dz); // top
    q({mn.x, mn.y, mn.z}, dx, dz);  // bottom
    return sides;
}

Translate::Translate(HittablePtr o, Vec3 off) : object(std::move(o)), offset(off) {
    if (!object || !object->bounding_box()) throw Error("Objects without bounding boxes should not be Translated");
    bbox = *object->bounding_box() + off;
}

RotateY::RotateY(HittablePtr o, double angle) : object(std::move(o)) {
    if (!object || !object->bounding_box()) throw Error("Objects without bounding boxes should not be Rotated");
    sin_theta = std::sin(angle);
    cos_theta = std::cos(angle);
    const BoundingBox3 &b = *object->bounding_box();
    double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                double x = i * b.x.end + (1 - i) * b.x.start;
                double y = j * b.y.end + (1 - j) * b.y.start;
                double z = k * b.z.end + (1 - k) * b.z.start;
                double t[3] = {cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z};
                for (int c = 0; c < 3; ++c) mn[c] = std::fmin(mn[c], t[c]), mx[c] = std::fmax(mx[c], t[c]);
            }
    bbox = BoundingBox3::bounded_by({mn[0], mn[1], mn[2]}, {mx[0], mx[1], mx[2]});
}

ConstantMedium::ConstantMedium(HittablePtr b, double density, TexturePtr tex)
    : boundary(std::move(b)), neg_inv_density(-1.0 / density), phase_fn(std::make_shared<Isotropic>(std::move(tex))) {
    if (!boundary) throw Error("ConstantMedium: null boundary");
}
std::shared_ptr<ConstantMedium> ConstantMedium::colored(HittablePtr b, double density, Color c) {
    return std::make_shared<ConstantMedium>(std::move(b), density, std::make_shared<SolidColor>(c));
}

BVHNode::BVHNode(std::vector<HittablePtr> objects) {
    // The reference has no arm for an empty list and recurses forever (boundingbox.rs:189-203).
    if (objects.empty()) throw Error("BVHNode::new: empty object list");
    for (auto &o : objects) {
        if (!o) throw Error("BVHNode::new: null object");
        const BoundingBox3 *ob = o->bounding_box();
        // extending_opt(Some(bbox), None) re-pads bbox and goes on; cmp_box would unwrap None later
        if (!ob) throw Error("BVHNode::new: object without bounding box");
        if (ob->x.start != ob->x.start || ob->y.start != ob->y.start || ob->z.start != ob->z.start)
            throw Error("Tried to cmp a NaN value"); // boundingbox.rs:214
        bbox = BoundingBox3::extending(bbox, *ob);
    }
    int axis = bbox.longest_axis();
    if (objects.size() == 1) {
        left = right = objects[0];
    } else if (objects.size() == 2) {
        left = objects[0];
        right = objects[1];
    } else {
        // sort_unstable_by on the interval start; the order of equal keys is unspecified there
        std::sort(objects.begin(), objects.end(), [axis](const HittablePtr &a, const HittablePtr &b) {
            return a->bounding_box()->axis(axis).start < b->bounding_box()->axis(axis).start;
        });
        size_t mid = objects.size() / 2;
        std::vector<HittablePtr> l(objects.begin(), objects.begin() + mid), r(objects.begin() + mid, objects.end());
        left = std::make_shared<BVHNode>(std::move(l));
        right = std::make_shared<BVHNode>(std::move(r));
    }
}
void BVHNode::stats(int depth, int &nodes, int &max_depth) const {
    nodes++;
    max_depth = std::max(max_depth, depth);
    if (left->kind() == BVH) static_cast<const BVHNode *>(left.get())->stats(depth + 1, nodes, max_depth);
    if (right != left && right->kind() == BVH) static_cast<const BVHNode *>(right.get())->stats(depth + 1, nodes, max_depth);
}

// ------------------------------------------------------------------ camera
CameraBuilder::CameraBuilder() { rth_camera_defaults(&b); }
static void add_err(std::vector<std::string> &e, bool cond, const std::string &msg) {
    if (cond) e.push_back("CameraBuilder::" + msg);
}
static std::string num(double v) {
    char buf[64];
    snprintf(buf, sizeof buf, "%g", v);
    return buf;
}
CameraBuilder &CameraBuilder::dimensions(uint32_t w, uint32_t h) {
    add_err(errors, w <= 1, "dimensions: Invalid width: must be greater than 1, found " + std::to_string(w));
    add_err(errors, h <= 1, "dimensions: Invalid height: must be greater than 1, found " + std::to_string(h));
    b.image_width = w, b.image_height = h;
    return *this;
}
CameraBuilder &CameraBuilder::with_aspect_ratio(uint32_t w, double ar) {
    add_err(errors, w <= 1, "with_aspect_ratio: Invalid width: must be greater than 1, found " + std::to_string(w));
    add_err(errors, !(ar >= 0.1 && ar <= 100.0),
            "with_aspect_ratio: Invalid aspect_ratio: must be between 0.1 and 100.0, found " + num(ar));
    b.image_width = w;
    double h = double(w) / ar;
    double r = std::round(h);
    b.image_height = (r != r || r <= 0.0) ? 0u : (r >= 4294967295.0 ? 4294967295u : uint32_t(r)); // saturating `as u32`
    return *this;
}
CameraBuilder &CameraBuilder::max_depth(uint32_t d) {
    add_err(errors, d < 1, "max_depth: Invalid depth: must be at least 1, found " + std::to_string(d));
    b.max_depth = d;
    return *this;
}
CameraBuilder &CameraBuilder::background(Background bg) {
    if (!bg.sky) add_err(errors, !bg.color.is_valid(), "background: Invalid color provided.");
    b.background_sky = bg.sky;
    b.background[0] = bg.color.r, b.background[1] = bg.color.g, b.background[2] = bg.color.b;
    return *this;
}
CameraBuilder &CameraBuilder::antialias(AntialiasingType type, uint32_t spp) {
    add_err(errors, spp < 1, "antialias: Invalid samples_per_px: must be at least 1, found " + std::to_string(spp));
    b.samples_per_px = spp;
    b.antialiasing = type == AntialiasingType::Disc ? int32_t(RT_AA_DISC) : int32_t(RT_AA_SQUARE);
    return *this;
}
CameraBuilder &CameraBuilder::camera_center(Point3 c) {
    b.center[0] = c.x, b.center[1] = c.y, b.center[2] = c.z;
    return *this;
}
CameraBuilder &CameraBuilder::focal_length(double l) {
    add_err(errors, l <= 0.0, "focal_length: Invalid length: must be greater than 0.0, found " + num(l));
    b.focal_length = l;
    return *this;
}
CameraBuilder &CameraBuilder::vfov(double deg) {
    add_err(errors, !(deg >= 0.01 && deg < 360.0), "vfov: Invalid deg: must be between 0.01 and 360.0, found " + num(deg));
    b.vfov_deg = deg;
    return *this;
}
CameraBuilder &CameraBuilder::camera_target(Point3 t) {
    b.target[0] = t.x, b.target[1] = t.y, b.target[2] = t.z;
    return *this;
}
CameraBuilder &CameraBuilder::vup(Vec3 v) {
    b.vup[0] = v.x, b.vup[1] = v.y, b.vup[2] = v.z;
    return *this;
}
CameraBuilder &CameraBuilder::defocus_angle(double a) {
    add_err(errors, !(a >= 0.0 && a < 180.0), "defocus_angle: Invalid angle: must be between 0.0 and 180.0, found " + num(a));
    b.defocus_angle_deg = a;
    return *this;
}
bool CameraBuilder::build(Camera &out, std::vector<std::string> &errs) const {
    if (!errors.empty()) {
        errs = errors;
        return false;
    }
    out = Camera::build(b);
    return true;
}

Camera Camera::build(const rth_camera_builder &b) { // camera.rs:280-366
    Camera cam;
    rt_camera &c = cam.derived;
    c.image_width = b.image_width;
    c.image_height = b.image_height;
    double fwidth = double(b.image_width), fheight = double(b.image_height);
    double aspect_ratio = fwidth / fheight;
    double vfov = b.vfov_deg * (PI / 180.0);
    double defocus_angle = b.defocus_angle_deg * (PI / 180.0);
    double focal_length = b.focal_length;
    Point3 center{b.center[0], b.center[1], b.center[2]}, target{b.target[0], b.target[1], b.target[2]};
    Vec3 vup = Vec3{b.vup[0], b.vup[1], b.vup[2]}.as_unit();
    double viewport_height = 2.0 * std::tan(vfov / 2.0) * focal_length;
    double viewport_width = viewport_height * aspect_ratio;
    Vec3 w = (center - target).as_unit();
    Vec3 u = Vec3::cross(vup, w); // labelled normalised by the reference, not renormalised (vec.rs:299-306)
    Vec3 v = Vec3::cross(w, u);
    Vec3 viewport_u = viewport_width * u;
    Vec3 viewport_v = viewport_height * (-v);
    Vec3 pxdelta_u = viewport_u / double(b.image_width);
    Vec3 pxdelta_v = viewport_v / double(b.image_height);
    Point3 corner = center - (focal_length * w) - viewport_u / 2.0 - viewport_v / 2.0;
    Point3 pixel_00 = corner + (pxdelta_u + pxdelta_v) / 2.0;
    double defocus_radius = focal_length * std::tan(defocus_angle / 2.0);
    Vec3 disk_u = u * defocus_radius, disk_v = v * defocus_radius;
    uint32_t sqrt_spp = uint32_t(std::sqrt(double(b.samples_per_px))); // camera.rs:341
    cam.samples_per_px = sqrt_spp * sqrt_spp;
    c.sqrt_spp = sqrt_spp;
    c.max_depth = b.max_depth;
    c.background_sky = b.background_sky ? 1u : 0u;
    c.antialiasing = b.antialiasing == int32_t(RT_AA_DISC) ? RT_AA_DISC : RT_AA_SQUARE;
    auto put = [](double *dst, Vec3 s) { dst[0] = s.x, dst[1] = s.y, dst[2] = s.z; };
    put(c.background, {b.background[0], b.background[1], b.background[2]});
    put(c.center, center);
    put(c.pixel_00, pixel_00);
    put(c.pxdelta_u, pxdelta_u);
    put(c.pxdelta_v, pxdelta_v);
    c.defocus_angle = defocus_angle;
    put(c.defocus_disk_u, disk_u);
    put(c.defocus_disk_v, disk_v);
    return cam;
}

} // namespace raytracing
