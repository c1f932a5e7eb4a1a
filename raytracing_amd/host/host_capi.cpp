// host_capi.cpp -- C view (rth_* in include/rt_amd.h) of the C++ host scene layer.
#include "raytracing.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace raytracing;

struct rth_scene {
    std::vector<TexturePtr> textures;
    std::vector<MaterialPtr> materials;
    std::vector<HittablePtr> objects;
    std::map<std::string, int> named_materials;
    HittablePtr world, lights;
    std::unique_ptr<FlatScene> flat;
    int nprims = 0;
    std::string err;
};

namespace {

Vec3 v3(const double *p) { return {p[0], p[1], p[2]}; }
int fail(rth_scene *s, const std::string &m) {
    if (s) s->err = m;
    return -1;
}
bool ok_tex(rth_scene *s, int t) { return t >= 0 && t < int(s->textures.size()); }
bool ok_mat(rth_scene *s, int m) { return m >= 0 && m < int(s->materials.size()); }
bool ok_obj(rth_scene *s, int o) { return o >= 0 && o < int(s->objects.size()); }
int push_tex(rth_scene *s, TexturePtr t) {
    s->textures.push_back(std::move(t));
    return int(s->textures.size()) - 1;
}
int push_mat(rth_scene *s, MaterialPtr m) {
    s->materials.push_back(std::move(m));
    return int(s->materials.size()) - 1;
}
int push_obj(rth_scene *s, HittablePtr o) {
    s->objects.push_back(std::move(o));
    return int(s->objects.size()) - 1;
}
// leaf primitives are numbered in creation order
void number_leaves(rth_scene *s, Hittable &h) {
    if (h.kind() == Hittable::LIST) {
        for (auto &o : static_cast<HittableVec &>(h).objects) number_leaves(s, *o);
    } else if (h.prim_id < 0) {
        h.prim_id = s->nprims++;
    }
}

template <typename F> int guarded(rth_scene *s, F &&f) {
    try {
        return f();
    } catch (const std::exception &e) {
        return fail(s, e.what());
    }
}

} // namespace

extern "C" {

void rth_camera_defaults(rth_camera_builder *b) { // camera.rs:183-202
    std::memset(b, 0, sizeof *b);
    b->image_width = 400;
    b->image_height = 200;
    b->vfov_deg = 90.0;
    b->samples_per_px = 10;
    b->max_depth = 10;
    b->background_sky = 1;
    b->target[2] = -1.0;
    b->vup[1] = 1.0;
    b->defocus_angle_deg = 0.0;
    b->focal_length = 1.0;
}

int rth_camera_with_aspect_ratio(rth_camera_builder *b, uint32_t width, double aspect_ratio) {
    CameraBuilder cb;
    cb.b = *b;
    cb.with_aspect_ratio(width, aspect_ratio);
    *b = cb.b;
    return cb.errors.empty() ? 0 : -1;
}

int rth_camera_build(const rth_camera_builder *b, rt_camera *out, char *err, uint32_t err_len) {
    // replay the builder's setters so every check of camera.rs:63-166 runs
    CameraBuilder cb;
    cb.dimensions(b->image_width, b->image_height)
        .max_depth(b->max_depth)
        .antialias(b->antialiasing == int32_t(RT_AA_DISC) ? AntialiasingType::Disc : AntialiasingType::Square, b->samples_per_px)
        .focal_length(b->focal_length)
        .vfov(b->vfov_deg)
        .defocus_angle(b->defocus_angle_deg)
        .camera_center(v3(b->center))
        .camera_target(v3(b->target))
        .vup(v3(b->vup));
    Background bg = b->background_sky ? Background::Sky() : Background::Constant({b->background[0], b->background[1], b->background[2]});
    cb.background(bg);
    Camera cam;
    std::vector<std::string> errors;
    if (!cb.build(cam, errors)) {
        std::string joined;
        for (auto &e : errors) joined += (joined.empty() ? "" : "\n") + e;
        if (err && err_len) {
            std::strncpy(err, joined.c_str(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return -1;
    }
    *out = cam.derived;
    if (err && err_len) err[0] = 0;
    return 0;
}

rth_scene *rth_scene_new(void) { return new rth_scene(); }
void rth_scene_free(rth_scene *s) { delete s; }
const char *rth_last_error(rth_scene *s) { return s ? s->err.c_str() : ""; }

int rth_tex_solid(rth_scene *s, double r, double g, double b) { return push_tex(s, std::make_shared<SolidColor>(Color{r, g, b})); }
int rth_tex_checker(rth_scene *s, double scale, int even, int odd) {
    if (!ok_tex(s, even) || !ok_tex(s, odd)) return fail(s, "checker: bad texture id");
    return push_tex(s, std::make_shared<Checkerboard>(scale, s->textures[even], s->textures[odd]));
}
int rth_tex_image(rth_scene *s, uint32_t w, uint32_t h, const uint8_t *rgb) {
    return guarded(s, [&] {
        if (!rgb || w == 0 || h == 0) throw Error("image: empty");
        return push_tex(s, std::make_shared<ImageTexture>(std::vector<uint8_t>(rgb, rgb + size_t(w) * h * 3), w, h));
    });
}
int rth_tex_image_png(rth_scene *s, const char *path) {
    return guarded(s, [&] { return push_tex(s, ImageTexture::load(path)); });
}
int rth_tex_noise(rth_scene *s, double scale, const double *ranvec, const int32_t *perm) {
    return guarded(s, [&] { return push_tex(s, std::make_shared<NoiseTexture>(scale, ranvec, perm)); });
}

int rth_mat_lambertian(rth_scene *s, int tex) {
    if (!ok_tex(s, tex)) return fail(s, "lambertian: bad texture id");
    return push_mat(s, std::make_shared<Lambertian>(s->textures[tex]));
}
int rth_mat_light(rth_scene *s, int tex) {
    if (!ok_tex(s, tex)) return fail(s, "light: bad texture id");
    return push_mat(s, std::make_shared<DiffuseLight>(s->textures[tex]));
}
int rth_mat_metal(rth_scene *s, double r, double g, double b, double fuzz) {
    return guarded(s, [&] { return push_mat(s, Metal::with_fuzz({r, g, b}, fuzz)); });
}
int rth_mat_dielectric(rth_scene *s, double ior) { return push_mat(s, std::make_shared<Dielectric>(ior)); }
int rth_mat_isotropic(rth_scene *s, int tex) {
    if (!ok_tex(s, tex)) return fail(s, "isotropic: bad texture id");
    return push_mat(s, std::make_shared<Isotropic>(s->textures[tex]));
}

int rth_sphere(rth_scene *s, const double c0[3], const double cv[3], double radius, int mat) {
    if (!ok_mat(s, mat)) return fail(s, "sphere: bad material id");
    return guarded(s, [&] {
        auto o = std::make_shared<Sphere>(v3(c0), v3(cv), radius, s->materials[mat]);
        o->prim_id = s->nprims++;
        return push_obj(s, o);
    });
}
int rth_quad(rth_scene *s, int shape, const double q[3], const double u[3], const double v[3], int mat) {
    if (!ok_mat(s, mat)) return fail(s, "quad: bad material id");
    if (shape < 0 || shape > 2) return fail(s, "quad: bad shape");
    return guarded(s, [&] {
        auto o = std::make_shared<Parallelogram>(v3(q), v3(u), v3(v), s->materials[mat], uint32_t(shape));
        o->prim_id = s->nprims++;
        return push_obj(s, o);
    });
}
int rth_box3(rth_scene *s, const double a[3], const double b[3], int mat) {
    if (!ok_mat(s, mat)) return fail(s, "box3: bad material id");
    return guarded(s, [&] {
        auto o = box3(v3(a), v3(b), s->materials[mat]);
        number_leaves(s, *o);
        return push_obj(s, o);
    });
}
int rth_list(rth_scene *s, const int *objs, int n) {
    return guarded(s, [&] {
        auto l = std::make_shared<HittableVec>();
        for (int i = 0; i < n; ++i) {
            if (!ok_obj(s, objs[i])) throw Error("list: bad object id");
            l->add(s->objects[objs[i]]);
        }
        return push_obj(s, l);
    });
}
int rth_list_add(rth_scene *s, int list, int obj) {
    if (!ok_obj(s, list) || !ok_obj(s, obj)) return fail(s, "list_add: bad object id");
    if (s->objects[list]->kind() != Hittable::LIST) return fail(s, "list_add: not a list");
    return guarded(s, [&] {
        static_cast<HittableVec &>(*s->objects[list]).add(s->objects[obj]);
        return 0;
    });
}
int rth_bvh(rth_scene *s, const int *objs, int n) {
    return guarded(s, [&] {
        std::vector<HittablePtr> v;
        for (int i = 0; i < n; ++i) {
            if (!ok_obj(s, objs[i])) throw Error("bvh: bad object id");
            v.push_back(s->objects[objs[i]]);
        }
        return push_obj(s, std::make_shared<BVHNode>(std::move(v)));
    });
}
int rth_translate(rth_scene *s, int obj, const double off[3]) {
    if (!ok_obj(s, obj)) return fail(s, "translate: bad object id");
    return guarded(s, [&] { return push_obj(s, std::make_shared<Translate>(s->objects[obj], v3(off))); });
}
int rth_rotate_y(rth_scene *s, int obj, double angle) {
    if (!ok_obj(s, obj)) return fail(s, "rotate_y: bad object id");
    return guarded(s, [&] { return push_obj(s, std::make_shared<RotateY>(s->objects[obj], angle)); });
}
int rth_medium(rth_scene *s, int boundary, double density, int tex) {
    if (!ok_obj(s, boundary)) return fail(s, "medium: bad object id");
    if (!ok_tex(s, tex)) return fail(s, "medium: bad texture id");
    return guarded(s, [&] {
        auto m = std::make_shared<ConstantMedium>(s->objects[boundary], density, s->textures[tex]);
        m->prim_id = s->nprims++;
        return push_obj(s, m);
    });
}
int rth_set_world(rth_scene *s, int obj) {
    if (!ok_obj(s, obj)) return fail(s, "set_world: bad object id");
    s->world = s->objects[obj];
    return 0;
}
int rth_set_lights(rth_scene *s, int obj) {
    if (obj < 0) {
        s->lights.reset();
        return 0;
    }
    if (!ok_obj(s, obj)) return fail(s, "set_lights: bad object id");
    s->lights = s->objects[obj];
    return 0;
}
int rth_prim_count(rth_scene *s) { return s->nprims; }

int rth_load_toml(rth_scene *s, const char *text) {
    return guarded(s, [&] {
        ConfigModel cfg = ConfigModel::from_str(text ? text : "");
        for (auto &kv : cfg.materials) s->named_materials[kv.first] = push_mat(s, kv.second);
        auto world = cfg.as_world();
        number_leaves(s, *world);
        for (auto &o : world->objects) push_obj(s, o);
        return push_obj(s, world);
    });
}
int rth_load_scene_toml(rth_scene *s, const char *text, const char *base_dir, rth_camera_builder *camera, int *has_camera) {
    return guarded(s, [&] {
        SceneFile f = load_scene_file(text ? text : "", base_dir ? base_dir : "", s->nprims);
        for (auto &kv : f.materials) s->named_materials[kv.first] = push_mat(s, kv.second);
        s->world = f.world;
        s->lights = f.lights;
        if (camera) *camera = f.camera;
        if (has_camera) *has_camera = f.has_camera ? 1 : 0;
        return push_obj(s, f.world);
    });
}
int rth_material_by_name(rth_scene *s, const char *name) {
    auto it = s->named_materials.find(name ? name : "");
    if (it == s->named_materials.end()) return fail(s, std::string("no material named ") + (name ? name : ""));
    return it->second;
}

int rth_flatten(rth_scene *s, const rt_scene_desc **out) {
    if (!s->world) return fail(s, "flatten: no world set");
    return guarded(s, [&] {
        s->flat = flatten(*s->world, s->lights.get());
        s->flat->desc.n_prims = std::max<uint32_t>(s->flat->desc.n_prims, uint32_t(s->nprims));
        *out = &s->flat->desc;
        return 0;
    });
}
int rth_bvh_stats(rth_scene *s, int obj, int *nodes, int *depth) {
    if (!ok_obj(s, obj) || s->objects[obj]->kind() != Hittable::BVH) return fail(s, "bvh_stats: not a BVH");
    *nodes = 0, *depth = 0;
    static_cast<const BVHNode &>(*s->objects[obj]).stats(1, *nodes, *depth);
    return 0;
}

void rth_color_hex(uint32_t hex, double rgb[3]) {
    Color c = Color::hex(hex);
    rgb[0] = c.r, rgb[1] = c.g, rgb[2] = c.b;
}
void rth_color_quantize(const double rgb[3], int gamma, uint8_t out[3]) {
    Color c{rgb[0], rgb[1], rgb[2]};
    if (gamma) c = c.as_gamma_corrected();
    auto q = c.as_rgb_ints();
    out[0] = q[0], out[1] = q[1], out[2] = q[2];
}
void rth_sphere_uv(const double p[3], double uv[2]) { // hittable.rs:150-162
    const double PI = 3.14159265358979323846264338327950288;
    double theta = std::acos(-p[1]);
    double phi = std::atan2(-p[2], p[0]) + PI;
    uv[0] = phi / (2.0 * PI);
    uv[1] = theta / PI;
}
int rth_interval_contains(double a, double b, double x) { return Interval{a, b}.contains(x); }
int rth_interval_surrounds(double a, double b, double x) { return Interval{a, b}.surrounds(x); }
int rth_interval_overlap(double a0, double a1, double b0, double b1, double out[2]) {
    auto r = Interval{a0, a1}.overlap(Interval{b0, b1});
    if (!r) return 0;
    out[0] = r->start, out[1] = r->end;
    return 1;
}
void rth_interval_expand(double a, double b, double delta, double out[2]) {
    Interval r = Interval{a, b}.expand(delta);
    out[0] = r.start, out[1] = r.end;
}
double rth_interval_clamp(double a, double b, double x) { return Interval{a, b}.clamp(x); }

int rth_write_png(const char *path, uint32_t w, uint32_t h, const float *rgb) {
    try {
        write_png(path, w, h, rgb);
        return 0;
    } catch (const std::exception &) {
        return -1;
    }
}
int rth_write_png_rgb8(const char *path, uint32_t w, uint32_t h, const uint8_t *rgb8) {
    try {
        write_png_rgb8(path, w, h, rgb8);
        return 0;
    } catch (const std::exception &) {
        return -1;
    }
}
int rth_write_ppm_rgb8(const char *path, uint32_t w, uint32_t h, const uint8_t *rgb8) {
    try {
        write_ppm_rgb8(path, w, h, rgb8);
        return 0;
    } catch (const std::exception &) {
        return -1;
    }
}
int rth_read_png_rgb8(const char *path, uint32_t *w, uint32_t *h, uint8_t *rgb8, uint64_t capacity) {
    try {
        std::vector<uint8_t> px;
        uint32_t ww = 0, hh = 0;
        read_png_rgb8(path, px, ww, hh);
        if (w) *w = ww;
        if (h) *h = hh;
        if (rgb8) {
            if (capacity < px.size()) return -2;
            std::memcpy(rgb8, px.data(), px.size());
        }
        return 0;
    } catch (const std::exception &) {
        return -1;
    }
}
int rth_write_ppm(const char *path, uint32_t w, uint32_t h, const float *rgb) {
    try {
        write_ppm(path, w, h, rgb);
        return 0;
    } catch (const std::exception &) {
        return -1;
    }
}

} // extern "C"
