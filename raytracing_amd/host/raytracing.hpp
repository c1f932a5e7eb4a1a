// raytracing.hpp -- host-side C++ mirror of the reference's scene / camera / config API.
//
// Same names and argument meaning as the Rust crate (src/lib.rs:13-41 re-exports), so a scene
// written against the reference reads the same here.  The objects carry DATA ONLY: there is no
// CPU intersection or shading code in this layer.  Rendering goes through flatten() and the
// device backend (include/rt_amd.h); without a GPU the render call fails.
#pragma once

#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

namespace raytracing {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ------------------------------------------------------------------ math (src/math/*.rs), f64
struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double len_squared() const { return x * x + y * y + z * z; }
    double len() const;
    Vec3 as_unit() const; // vec.rs:259-272
    static double dot(const Vec3 &a, const Vec3 &b);  // mul_add chain, vec.rs:122-124
    static Vec3 cross(const Vec3 &a, const Vec3 &b);  // mul_add form, vec.rs:163-170
};
using Point3 = Vec3;
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return a + (-b); }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(double s, Vec3 a) { return a * s; }
inline Vec3 operator/(Vec3 a, double s) { return a * (1.0 / s); } // vec.rs:484-490

struct Color { // math/color.rs
    double r = 0, g = 0, b = 0;
    Color() = default;
    Color(double r_, double g_, double b_) : r(r_), g(g_), b(b_) {}
    static Color new_ints(uint8_t r, uint8_t g, uint8_t b) { return {r / 255.0, g / 255.0, b / 255.0}; }
    static Color hex(uint32_t color); // color.rs:46-64 (<= 0xfff is three nibbles, each * 16)
    static Color black() { return {0, 0, 0}; }
    static Color white() { return {1, 1, 1}; }
    static Color over_white(double n) { return {n, n, n}; }
    bool is_valid() const; // color.rs:158-162
    void set_brightness(double k) { r *= k, g *= k, b *= k; }
    std::array<uint8_t, 3> as_rgb_ints() const; // clamp, *255, truncate (color.rs:148-155)
    Color as_gamma_corrected() const;           // sqrt, 0 if <= 0 (color.rs:130-136, 209-215)
};

struct Interval { // math/interval.rs
    double start = 0, end = 0;
    Interval() = default;
    Interval(double s, double e) : start(s), end(e) {}
    static Interval positive(double a, double b) { return a <= b ? Interval{a, b} : Interval{b, a}; }
    static Interval empty();
    static Interval universe();
    static Interval surrounding(const Interval &a, const Interval &b);
    std::optional<Interval> overlap(const Interval &o) const; // None when size <= 0
    double size() const { return end - start; }
    bool contains(double x) const { return start <= x && x <= end; }
    bool surrounds(double x) const { return start < x && x < end; }
    double clamp(double x) const { return x < start ? start : (x > end ? end : x); }
    Interval expand(double delta) const { return {start - delta / 2.0, end + delta / 2.0}; }
};

class BoundingBox3 { // boundingbox.rs:4-169
  public:
    Interval x, y, z;
    BoundingBox3() : x(Interval::empty()), y(Interval::empty()), z(Interval::empty()) {}
    BoundingBox3(Interval x, Interval y, Interval z); // pads axes thinner than 1e-4
    static BoundingBox3 bounded_by(const Point3 &a, const Point3 &b);
    static BoundingBox3 extending(const BoundingBox3 &a, const BoundingBox3 &b);
    const Interval &axis(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    int longest_axis() const; // strict >, ties to the earlier axis, empty -> X
    BoundingBox3 operator+(const Vec3 &off) const;
};

// ------------------------------------------------------------------ textures (texture.rs)
class Texture {
  public:
    enum Kind { SOLID, CHECKER, IMAGE, NOISE };
    virtual ~Texture() = default;
    virtual Kind kind() const = 0;
};
using TexturePtr = std::shared_ptr<Texture>;

class SolidColor : public Texture {
  public:
    Color albedo;
    explicit SolidColor(Color c) : albedo(c) {}
    Kind kind() const override { return SOLID; }
};
class Checkerboard : public Texture {
  public:
    double scale;
    TexturePtr even, odd;
    Checkerboard(double s, TexturePtr e, TexturePtr o) : scale(s), even(std::move(e)), odd(std::move(o)) {}
    static std::shared_ptr<Checkerboard> solid(double s, Color even, Color odd);
    Kind kind() const override { return CHECKER; }
};
class ImageTexture : public Texture {
  public:
    std::vector<uint8_t> image_data; // RGB8, row major
    uint32_t width, height;
    ImageTexture(std::vector<uint8_t> data, uint32_t w, uint32_t h); // asserts len == w*h*3
    static std::shared_ptr<ImageTexture> load(const std::string &png_path); // 8-bit RGB only
    Kind kind() const override { return IMAGE; }
};
// Perlin marble of "The Next Week"; the reference has no noise texture.
class NoiseTexture : public Texture {
  public:
    double scale;
    std::array<Vec3, 256> ranvec;
    std::array<std::array<int, 256>, 3> perm;
    NoiseTexture(double scale, uint64_t seed);
    NoiseTexture(double scale, const double *ranvec768, const int32_t *perm768);
    Kind kind() const override { return NOISE; }
};

// ------------------------------------------------------------------ materials (material.rs)
class Material {
  public:
    enum Kind { LAMBERTIAN, LIGHT, METAL, DIELECTRIC, ISOTROPIC };
    virtual ~Material() = default;
    virtual Kind kind() const = 0;
};
using MaterialPtr = std::shared_ptr<Material>;

class Lambertian : public Material {
  public:
    TexturePtr texture;
    explicit Lambertian(TexturePtr t) : texture(std::move(t)) {}
    static std::shared_ptr<Lambertian> solid(Color albedo);
    Kind kind() const override { return LAMBERTIAN; }
};
class DiffuseLight : public Material {
  public:
    TexturePtr texture;
    explicit DiffuseLight(TexturePtr t) : texture(std::move(t)) {}
    static std::shared_ptr<DiffuseLight> solid(Color albedo);
    Kind kind() const override { return LIGHT; }
};
class Metal : public Material {
  public:
    Color albedo;
    double fuzz = 0.0;
    explicit Metal(Color a) : albedo(a) {}
    static std::shared_ptr<Metal> with_fuzz(Color albedo, double fuzz); // throws unless 0..=1
    Kind kind() const override { return METAL; }
};
class Dielectric : public Material {
  public:
    double refraction_index;
    explicit Dielectric(double ri) : refraction_index(ri) {}
    Kind kind() const override { return DIELECTRIC; }
};
class Isotropic : public Material {
  public:
    TexturePtr texture;
    explicit Isotropic(TexturePtr t) : texture(std::move(t)) {}
    static std::shared_ptr<Isotropic> colored(Color c);
    Kind kind() const override { return ISOTROPIC; }
};

// ------------------------------------------------------------------ hittables (hittable.rs)
class Hittable {
  public:
    enum Kind { SPHERE, QUAD, LIST, TRANSLATE, ROTATE_Y, MEDIUM, BVH };
    virtual ~Hittable() = default;
    virtual Kind kind() const = 0;
    virtual const BoundingBox3 *bounding_box() const = 0;
    int prim_id = -1; // leaves only; -1 = numbered by flatten() in depth-first order
};
using HittablePtr = std::shared_ptr<Hittable>;

class Sphere : public Hittable { // centre moves along Ray3 {origin, direction} over time 0..1
  public:
    Point3 center0;
    Vec3 center_vec;
    double radius;
    MaterialPtr material;
    BoundingBox3 bbox;
    Sphere(Point3 c0, Vec3 cv, double r, MaterialPtr m); // hittable.rs:130-148; throws if r < 0
    static std::shared_ptr<Sphere> stationary(Point3 c, double r, MaterialPtr m);
    Kind kind() const override { return SPHERE; }
    const BoundingBox3 *bounding_box() const override { return &bbox; }
};

class Parallelogram : public Hittable { // also Triangle and Disc (same plane set-up)
  public:
    uint32_t shape;
    Point3 corner;
    Vec3 u, v;
    MaterialPtr material;
    BoundingBox3 bbox;
    Parallelogram(Point3 corner, Vec3 u, Vec3 v, MaterialPtr m, uint32_t shape = RT_SHAPE_PARALLELOGRAM);
    Kind kind() const override { return QUAD; }
    const BoundingBox3 *bounding_box() const override { return &bbox; }
};
std::shared_ptr<Parallelogram> Triangle(Point3 corner, Vec3 u, Vec3 v, MaterialPtr m);
std::shared_ptr<Parallelogram> TriangleFromPoints(Point3 a, Point3 b, Point3 c, MaterialPtr m);
std::shared_ptr<Parallelogram> Disc(Point3 corner, Vec3 u, Vec3 v, MaterialPtr m);
std::shared_ptr<Parallelogram> DiscFromCenter(Point3 center, Vec3 u, Vec3 v, MaterialPtr m); // :596-619

class HittableVec : public Hittable {
  public:
    std::vector<HittablePtr> objects;
    std::optional<BoundingBox3> bbox;
    void add(HittablePtr obj); // hittable.rs:238-244; throws if obj has no bounding box
    size_t len() const { return objects.size(); }
    Kind kind() const override { return LIST; }
    const BoundingBox3 *bounding_box() const override { return bbox ? &*bbox : nullptr; }
};
std::shared_ptr<HittableVec> box3(const Point3 &a, const Point3 &b, MaterialPtr m); // hittable.rs:398-455

class Translate : public Hittable {
  public:
    HittablePtr object;
    Vec3 offset;
    BoundingBox3 bbox;
    Translate(HittablePtr o, Vec3 off);
    Kind kind() const override { return TRANSLATE; }
    const BoundingBox3 *bounding_box() const override { return &bbox; }
};
class RotateY : public Hittable {
  public:
    HittablePtr object;
    double sin_theta, cos_theta;
    BoundingBox3 bbox;
    RotateY(HittablePtr o, double angle_rad); // hittable.rs:727-763
    Kind kind() const override { return ROTATE_Y; }
    const BoundingBox3 *bounding_box() const override { return &bbox; }
};
class ConstantMedium : public Hittable {
  public:
    HittablePtr boundary;
    double neg_inv_density;
    std::shared_ptr<Isotropic> phase_fn;
    ConstantMedium(HittablePtr boundary, double density, TexturePtr tex);
    static std::shared_ptr<ConstantMedium> colored(HittablePtr boundary, double density, Color c);
    Kind kind() const override { return MEDIUM; }
    const BoundingBox3 *bounding_box() const override { return boundary->bounding_box(); }
};
class BVHNode : public Hittable { // boundingbox.rs:171-236 -- built on the host
  public:
    HittablePtr left, right;
    BoundingBox3 bbox;
    explicit BVHNode(std::vector<HittablePtr> objects); // throws on an empty list
    Kind kind() const override { return BVH; }
    const BoundingBox3 *bounding_box() const override { return &bbox; }
    void stats(int depth, int &nodes, int &max_depth) const;
};

// ------------------------------------------------------------------ flattening
struct FlatScene {
    std::vector<rt_node> nodes;
    std::vector<uint32_t> refs;
    std::vector<rt_quad> quads;
    std::vector<rt_sphere> spheres;
    std::vector<rt_medium> media;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<uint32_t> light_refs;
    std::vector<uint8_t> texels;
    std::vector<float> noise_vec;
    std::vector<int32_t> noise_perm;
    rt_scene_desc desc{};
    void finish(); // points desc at the vectors
};
// Bakes Translate/RotateY into the primitives (f64), linearises lists and BVH nodes breadth
// first, converts to fp32 with outward-rounded boxes.  lights may be null.
std::unique_ptr<FlatScene> flatten(const Hittable &world, const Hittable *lights);

// ------------------------------------------------------------------ camera (camera.rs)
enum class AntialiasingType { Square, Disc }; // stored, never read by the reference (camera.rs:237)
struct Background {
    bool sky = true;
    Color color;
    static Background Sky() { return {}; }
    static Background Constant(Color c) { return {false, c}; }
};

class Camera;
class CameraBuilder { // camera.rs:14-202
  public:
    CameraBuilder();
    CameraBuilder &dimensions(uint32_t width, uint32_t height);
    CameraBuilder &with_aspect_ratio(uint32_t width, double aspect_ratio);
    CameraBuilder &max_depth(uint32_t depth);
    CameraBuilder &background(Background bg);
    CameraBuilder &antialias(AntialiasingType type, uint32_t samples_per_px);
    CameraBuilder &camera_center(Point3 c);
    CameraBuilder &focal_length(double len);
    CameraBuilder &vfov(double deg);
    CameraBuilder &camera_target(Point3 t);
    CameraBuilder &vup(Vec3 v);
    CameraBuilder &defocus_angle(double deg);
    // Ok(Camera) or the list of "CameraBuilder::..." messages (camera.rs:173-180).  The
    // reference also demands a writer here; this mirror returns the film to the caller instead.
    bool build(Camera &out, std::vector<std::string> &errors) const;

    rth_camera_builder b{};
    std::vector<std::string> errors;
};

// export.rs:5-8.  write() receives the whole frame, row-major, already scaled by 1/spp.
class ImageWriter {
  public:
    virtual ~ImageWriter() = default;
    virtual void write_header(uint32_t width, uint32_t height) = 0;
    virtual void write(const std::vector<Color> &buf) = 0;
};
class PpmWriter : public ImageWriter { // export.rs:10-41: P3 text, gamma 2
  public:
    explicit PpmWriter(std::string path) : path_(std::move(path)) {}
    void write_header(uint32_t width, uint32_t height) override { w_ = width, h_ = height; }
    void write(const std::vector<Color> &buf) override;

  private:
    std::string path_;
    uint32_t w_ = 0, h_ = 0;
};
class PngWriter : public ImageWriter { // export.rs:43-86: 8-bit RGB, linear values
  public:
    explicit PngWriter(std::string path) : path_(std::move(path)) {}
    void write_header(uint32_t width, uint32_t height) override { w_ = width, h_ = height; }
    void write(const std::vector<Color> &buf) override;

  private:
    std::string path_;
    uint32_t w_ = 0, h_ = 0;
};

// What the device needs beyond the reference's render(world, lights): which GPU, which
// estimator (the shipped one by default), and the seed the reference draws from the OS.
struct RenderSettings {
    int device_id = 0;
    // More than one entry: the frame's strata (camera.rs:390-395) are split over these GPUs, every one
    // renders the full frame for its own range, and the films are summed by rt_film_allreduce (RCCL);
    // device_id is then ignored.  One process, one handle and one host thread per GPU.
    std::vector<int> devices;
    uint32_t mode = RT_MODE_REF_EXACT;
    uint32_t flags = 0; // RT_FLAG_*
    uint64_t seed = 1;
};
struct RenderReport {
    uint64_t segments = 0, camera_rays = 0, dropped = 0;
    double render_ms = 0.0;
};

class Camera {
  public:
    rt_camera derived{};      // everything Camera::build computes (camera.rs:280-366)
    uint32_t samples_per_px;  // sqrt_spp^2
    static Camera build(const rth_camera_builder &b);
    // Camera::render (camera.rs:368-404) on the device: flattens the scene graph, uploads it,
    // traces every stratum of every pixel and hands the frame (sums x 1/spp, camera.rs:397) to
    // the writer.  lights may be null for a scene without sampled lights (RT_MODE_BOOK3 only).
    // Throws Error with the device's message; there is no CPU rendering path.
    void render(const Hittable &world, const Hittable *lights, ImageWriter &writer, const RenderSettings &settings = {},
                RenderReport *report = nullptr) const;
    std::vector<Color> render(const Hittable &world, const Hittable *lights, const RenderSettings &settings = {},
                              RenderReport *report = nullptr) const;
};

// ------------------------------------------------------------------ config (config.rs)
class ConfigModel {
  public:
    static ConfigModel from_str(const std::string &toml_text); // throws Error with the message
    std::shared_ptr<HittableVec> as_world() const;
    std::map<std::string, MaterialPtr> materials;
    std::vector<HittablePtr> objects;
};

// A scene file: ConfigModel's format plus what a whole render needs (see config.cpp).
struct SceneFile {
    std::shared_ptr<HittableVec> world;
    HittablePtr lights; // null when no object is marked `light = true`
    std::map<std::string, MaterialPtr> materials;
    bool has_camera = false;
    rth_camera_builder camera; // defaults of camera.rs:183-202 overridden by [camera]
};
// Primitives are numbered from next_prim_id in the order the file lists them.
SceneFile load_scene_file(const std::string &toml_text, const std::string &base_dir, int &next_prim_id);

// ------------------------------------------------------------------ image writers (export.rs)
void write_png(const std::string &path, uint32_t w, uint32_t h, const float *rgb); // linear, truncating
void write_ppm(const std::string &path, uint32_t w, uint32_t h, const float *rgb); // gamma 2, truncating
void write_png_rgb8(const std::string &path, uint32_t w, uint32_t h, const uint8_t *rgb8); // bytes as given
void write_ppm_rgb8(const std::string &path, uint32_t w, uint32_t h, const uint8_t *rgb8);
void read_png_rgb8(const std::string &path, std::vector<uint8_t> &rgb, uint32_t &w, uint32_t &h);

} // namespace raytracing
