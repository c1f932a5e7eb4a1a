// main.cpp -- the counterpart of the reference's binary (main.rs:12-72) on the device backend.
//
//   rt_main [--config cornell_box.toml] [--out image.png] [--spp 20] [--width 600] [--seed 1] [--gpus N]
//   rt_main --scene scenes/book1_final/book1_final.toml [--out image.png] [--spp N] [--width W]
//           [--height H] [--mode ref-exact|book3] [--ppm]
//
// Without --scene it renders what main.rs renders: the walls of cornell_box.toml plus the light
// quad and the two rotated boxes added in code, with the shipped estimator.  With --scene the
// file describes the whole render (camera, lights, instances); the mixture estimator is the
// default there because most such scenes hold materials the shipped one cannot scatter.
// --gpus N splits the strata over GPUs 0 .. N-1 of this process and sums the films with
// rt_film_allreduce (RCCL); --device picks the GPU of a one-GPU render.
#include "raytracing.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

using namespace raytracing;

namespace {

std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("cannot read " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::string dir_of(const std::string &path) {
    size_t k = path.find_last_of('/');
    return k == std::string::npos ? std::string(".") : path.substr(0, k);
}

double radians(double deg) { return deg * M_PI / 180.0; }

struct Args {
    std::string config = "cornell_box.toml", scene, out = "image.png";
    long spp = -1, width = -1, height = -1, device = 0, gpus = 1;
    unsigned long long seed = 1;
    int mode = -1;
    bool ppm = false;
};

Args parse(int argc, char **argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) throw Error("missing value after " + k);
            return argv[++i];
        };
        if (k == "--config") a.config = val();
        else if (k == "--scene") a.scene = val();
        else if (k == "--out") a.out = val();
        else if (k == "--spp") a.spp = std::atol(val());
        else if (k == "--width") a.width = std::atol(val());
        else if (k == "--height") a.height = std::atol(val());
        else if (k == "--device") a.device = std::atol(val());
        else if (k == "--gpus") a.gpus = std::atol(val());
        else if (k == "--seed") a.seed = std::strtoull(val(), nullptr, 10);
        else if (k == "--ppm") a.ppm = true;
        else if (k == "--mode") {
            std::string m = val();
            if (m == "ref-exact") a.mode = RT_MODE_REF_EXACT;
            else if (m == "book3") a.mode = RT_MODE_BOOK3;
            else throw Error("--mode takes ref-exact or book3");
        } else
            throw Error("unknown argument " + k);
    }
    return a;
}

} // namespace

int main(int argc, char **argv) {
    try {
        Args a = parse(argc, argv);
        std::shared_ptr<HittableVec> world;
        HittablePtr lights;
        CameraBuilder cb;
        RenderSettings rs;
        rs.device_id = int(a.device);
        rs.seed = a.seed;
        if (a.gpus < 1 || a.gpus > 64) throw Error("--gpus takes 1 to 64");
        if (a.gpus > 1)
            for (long g = 0; g < a.gpus; ++g) rs.devices.push_back(int(g));
        if (a.scene.empty()) {
            // main.rs:15-26
            cb.with_aspect_ratio(600, 1.0)
                .max_depth(50)
                .antialias(AntialiasingType::Square, 20)
                .background(Background::Constant(Color::black()))
                .camera_center(Point3(278.0, 278.0, -800.0))
                .camera_target(Point3(278.0, 278.0, 0.0))
                .vfov(40.0)
                .defocus_angle(0.0);
            // main.rs:28-30
            world = ConfigModel::from_str(slurp(a.config)).as_world();
            // main.rs:32-51
            MaterialPtr white = Lambertian::solid(Color::white());
            Color light_color = Color::white();
            light_color.set_brightness(20.0);
            MaterialPtr light = DiffuseLight::solid(light_color);
            auto lightbox = std::make_shared<Parallelogram>(Point3(343.0, 554.0, 332.0), Vec3(-130.0, 0.0, 0.0), Vec3(0.0, 0.0, -105.0), light);
            world->add(lightbox);
            // main.rs:53-69
            HittablePtr box1 = box3(Point3(0, 0, 0), Point3(165.0, 330.0, 165.0), white);
            box1 = std::make_shared<RotateY>(box1, radians(15.0));
            box1 = std::make_shared<Translate>(box1, Vec3(265.0, 0.0, 295.0));
            world->add(box1);
            HittablePtr box2 = box3(Point3(0, 0, 0), Point3(165.0, 165.0, 165.0), white);
            box2 = std::make_shared<RotateY>(box2, radians(-18.0));
            box2 = std::make_shared<Translate>(box2, Vec3(130.0, 0.0, 65.0));
            world->add(box2);
            lights = lightbox;
            rs.mode = a.mode >= 0 ? uint32_t(a.mode) : RT_MODE_REF_EXACT;
        } else {
            int next_id = 0;
            SceneFile sf = load_scene_file(slurp(a.scene), dir_of(a.scene), next_id);
            if (!sf.has_camera) throw Error(a.scene + ": no [camera] table");
            world = sf.world;
            lights = sf.lights;
            cb.b = sf.camera;
            rs.mode = a.mode >= 0 ? uint32_t(a.mode) : RT_MODE_BOOK3;
        }
        if (a.width > 0 && a.height > 0) cb.dimensions(uint32_t(a.width), uint32_t(a.height));
        else if (a.width > 0) cb.dimensions(uint32_t(a.width), uint32_t(std::llround(double(a.width) * cb.b.image_height / cb.b.image_width)));
        if (a.spp > 0) cb.b.samples_per_px = uint32_t(a.spp);
        Camera cam;
        std::vector<std::string> errors;
        if (!cb.build(cam, errors)) { // camera.rs:173-180
            for (const std::string &e : errors) std::fprintf(stderr, "%s\n", e.c_str());
            return 2;
        }
        RenderReport rep;
        if (a.ppm) {
            PpmWriter w(a.out);
            cam.render(*world, lights.get(), w, rs, &rep);
        } else {
            PngWriter w(a.out);
            cam.render(*world, lights.get(), w, rs, &rep);
        }
        std::fprintf(stderr, "%ux%u, %u spp: %llu segments in %.2f ms (%.1f M segments/s) -> %s\n", cam.derived.image_width,
                     cam.derived.image_height, cam.samples_per_px, (unsigned long long)rep.segments, rep.render_ms,
                     rep.render_ms > 0 ? rep.segments / rep.render_ms / 1e3 : 0.0, a.out.c_str());
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
