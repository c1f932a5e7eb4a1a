// render.cpp -- Camera::render (camera.rs:368-404) and the ImageWriters (export.rs) of the
// C++ host layer: the scene graph goes through flatten() to the device backend (rt_* calls of
// include/rt_amd.h, same library); nothing is intersected or shaded on the CPU.
#include "raytracing.hpp"

namespace raytracing {
namespace {

struct Handle { // rt_destroy on every path out
    rt_handle *h = nullptr;
    ~Handle() {
        if (h) rt_destroy(h);
    }
};

void check(rt_handle *h, int rc, const char *what) {
    if (rc != 0) throw Error(std::string(what) + ": " + (h ? rt_last_error(h) : "no device handle"));
}

std::vector<uint8_t> to_bytes(const std::vector<Color> &buf, size_t n, bool gamma) {
    if (buf.size() != n) throw Error("ImageWriter::write: buffer does not match the header's dimensions");
    std::vector<uint8_t> px(3 * n);
    for (size_t i = 0; i < n; ++i) {
        auto q = (gamma ? buf[i].as_gamma_corrected() : buf[i]).as_rgb_ints();
        px[3 * i] = q[0], px[3 * i + 1] = q[1], px[3 * i + 2] = q[2];
    }
    return px;
}

} // namespace

void PpmWriter::write(const std::vector<Color> &buf) { // export.rs:28-41
    write_ppm_rgb8(path_, w_, h_, to_bytes(buf, size_t(w_) * h_, true).data());
}

void PngWriter::write(const std::vector<Color> &buf) { // export.rs:63-86
    write_png_rgb8(path_, w_, h_, to_bytes(buf, size_t(w_) * h_, false).data());
}

std::vector<Color> Camera::render(const Hittable &world, const Hittable *lights, const RenderSettings &settings,
                                  RenderReport *report) const {
    std::unique_ptr<FlatScene> flat = flatten(world, lights);
    Handle dev;
    int rc = rt_create(settings.device_id, &dev.h);
    if (rc != 0) throw Error(std::string("rt_create: ") + rt_last_error(nullptr));
    check(dev.h, rt_upload_scene(dev.h, &flat->desc), "rt_upload_scene");
    check(dev.h, rt_film_resize(dev.h, derived.image_width, derived.image_height), "rt_film_resize");
    rt_render_params p{};
    p.struct_size = sizeof(p);
    p.mode = settings.mode;
    p.flags = settings.flags;
    p.seed = settings.seed;
    check(dev.h, rt_render(dev.h, &derived, &p), "rt_render"); // all sqrt_spp^2 strata
    rt_stats st{};
    st.struct_size = sizeof(st);
    check(dev.h, rt_get_stats(dev.h, &st), "rt_get_stats");
    if (report) report->segments = st.segments, report->camera_rays = st.camera_rays, report->dropped = st.dropped, report->render_ms = st.render_ms;
    const size_t n = size_t(derived.image_width) * derived.image_height;
    std::vector<float> sums(3 * n);
    check(dev.h, rt_film_download(dev.h, sums.data(), 1.0f), "rt_film_download");
    // px_sample_scale in f64 on the fp32 sums (camera.rs:397): the same products the device's
    // own 8-bit stage forms, so both routes give the same bytes
    const double scale = 1.0 / double(samples_per_px);
    std::vector<Color> buf(n);
    for (size_t i = 0; i < n; ++i) buf[i] = Color(double(sums[3 * i]) * scale, double(sums[3 * i + 1]) * scale, double(sums[3 * i + 2]) * scale);
    return buf;
}

void Camera::render(const Hittable &world, const Hittable *lights, ImageWriter &writer, const RenderSettings &settings,
                    RenderReport *report) const {
    writer.write_header(derived.image_width, derived.image_height); // camera.rs:369-371
    writer.write(render(world, lights, settings, report));          // camera.rs:403
}

} // namespace raytracing
