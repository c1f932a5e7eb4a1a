// render.cpp -- Camera::render (camera.rs:368-404) and the ImageWriters (export.rs) of the
// C++ host layer: the scene graph goes through flatten() to the device backend (rt_* calls of
// include/rt_amd.h, same library); nothing is intersected or shaded on the CPU.
#include "raytracing.hpp"

#include <thread>

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

namespace {

// Contiguous strata [begin, end) of `rank` among `n`: the first n_strata % n ranks get one more.
void strata_of(uint32_t n_strata, uint32_t n, uint32_t rank, uint32_t &begin, uint32_t &end) {
    const uint32_t base = n_strata / n, rem = n_strata % n;
    begin = rank * base + std::min(rank, rem);
    end = begin + base + (rank < rem ? 1u : 0u);
}

// The frame on several GPUs of this process: the same samples as on one (a sample's random numbers are
// keyed by pixel and stratum), summed in another order.
std::vector<float> render_on_devices(const rt_camera &derived, uint32_t samples_per_px, const rt_scene_desc &desc,
                                     const RenderSettings &settings, RenderReport *report) {
    const size_t n = settings.devices.size();
    std::vector<Handle> devs(n);
    std::vector<rt_handle *> hs(n, nullptr);
    for (size_t i = 0; i < n; ++i) {
        if (rt_create(settings.devices[i], &devs[i].h) != 0) throw Error(std::string("rt_create: ") + rt_last_error(nullptr));
        hs[i] = devs[i].h;
        check(hs[i], rt_upload_scene(hs[i], &desc), "rt_upload_scene");
        check(hs[i], rt_film_resize(hs[i], derived.image_width, derived.image_height), "rt_film_resize");
    }
    check(hs[0], rt_comm_init_all(hs.data(), int(n)), "rt_comm_init_all");
    std::vector<std::string> errors(n);
    std::vector<rt_stats> stats(n);
    std::vector<std::thread> threads;
    for (size_t i = 0; i < n; ++i)
        threads.emplace_back([&, i]() { // rt_render returns when the GPU has finished: one host thread per GPU
            rt_render_params p{};
            p.struct_size = sizeof(p);
            p.mode = settings.mode, p.flags = settings.flags, p.seed = settings.seed;
            strata_of(samples_per_px, uint32_t(n), uint32_t(i), p.stratum_begin, p.stratum_end);
            stats[i] = rt_stats{};
            stats[i].struct_size = sizeof(rt_stats);
            if (p.stratum_end == p.stratum_begin) return; // more GPUs than strata: this one adds nothing
            if (rt_render(hs[i], &derived, &p) != 0 || rt_get_stats(hs[i], &stats[i]) != 0) errors[i] = rt_last_error(hs[i]);
        });
    for (std::thread &t : threads) t.join();
    for (size_t i = 0; i < n; ++i)
        if (!errors[i].empty()) throw Error("rt_render on GPU " + std::to_string(settings.devices[i]) + ": " + errors[i]);
    check(hs[0], rt_film_allreduce(hs.data(), int(n)), "rt_film_allreduce");
    if (report) {
        *report = RenderReport{};
        for (const rt_stats &st : stats) {
            report->segments += st.segments, report->camera_rays += st.camera_rays, report->dropped += st.dropped;
            report->render_ms = std::max(report->render_ms, st.render_ms);
        }
    }
    std::vector<float> sums(3 * size_t(derived.image_width) * derived.image_height);
    check(hs[0], rt_film_download(hs[0], sums.data(), 1.0f), "rt_film_download"); // on the handle's stream, behind its reduce
    for (size_t i = 1; i < n; ++i) check(hs[i], rt_synchronize(hs[i]), "rt_synchronize"); // their reduces, before the handles go
    return sums;
}

} // namespace

std::vector<Color> Camera::render(const Hittable &world, const Hittable *lights, const RenderSettings &settings,
                                  RenderReport *report) const {
    std::unique_ptr<FlatScene> flat = flatten(world, lights);
    if (settings.devices.size() > 1) {
        const std::vector<float> sums = render_on_devices(derived, samples_per_px, flat->desc, settings, report);
        const double scale = 1.0 / double(samples_per_px);
        std::vector<Color> buf(sums.size() / 3);
        for (size_t i = 0; i < buf.size(); ++i) buf[i] = Color(double(sums[3 * i]) * scale, double(sums[3 * i + 1]) * scale, double(sums[3 * i + 2]) * scale);
        return buf;
    }
    Handle dev;
    int rc = rt_create(settings.devices.size() == 1 ? settings.devices[0] : settings.device_id, &dev.h);
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
