// host_sanitize.cpp -- the host layer under AddressSanitizer + UBSan (no GPU, no HIP).
// Built by `make sanitize` from the host sources themselves; run by tests/test_host_layer.py.
//   1. every scene file given on the command line: load, flatten, build the 4-wide BVH with both
//      builders, verify it (every primitive reachable, child boxes inside parent planes);
//   2. mutation fuzzing of the TOML loaders (deterministic): a damaged file must end in an error
//      message or a valid scene, never in a crash or a sanitizer report;
//   3. PNG writer -> reader round trip and the reader on damaged files;
//   4. camera builder on hostile numbers.
#include "../../raytracing_amd/csrc/rt_accel.h"
#include "../../raytracing_amd/csrc/rt_flat.h"
#include "../../raytracing_amd/host/raytracing.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

uint64_t rng_state = 0x9E3779B97F4A7C15ull;
uint32_t rnd() { // xorshift64*
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return uint32_t((rng_state * 0x2545F4914F6CDD1Dull) >> 32);
}

std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        std::fprintf(stderr, "cannot read %s\n", path.c_str());
        std::exit(2);
    }
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
std::string dir_of(const std::string &path) {
    size_t k = path.find_last_of('/');
    return k == std::string::npos ? std::string(".") : path.substr(0, k);
}

int failures = 0;
void check(bool ok, const std::string &what) {
    if (!ok) {
        std::fprintf(stderr, "FAIL: %s\n", what.c_str());
        ++failures;
    }
}

// returns true if the text loaded and flattened
bool load_and_build(const std::string &text, const std::string &base, bool scene_format, bool verify, size_t *n_prims = nullptr) {
    rth_scene *s = rth_scene_new();
    bool ok = false;
    rth_camera_builder cam;
    int has_camera = 0;
    int world = scene_format ? rth_load_scene_toml(s, text.c_str(), base.c_str(), &cam, &has_camera) : rth_load_toml(s, text.c_str());
    if (world >= 0) {
        if (!scene_format) rth_set_world(s, world);
        const rt_scene_desc *desc = nullptr;
        if (rth_flatten(s, &desc) == 0 && desc) {
            ok = true;
            if (n_prims) *n_prims = size_t(desc->n_quads) + desc->n_spheres + desc->n_media;
            {
                // the flat scan's records: whatever the list holds, the plan accounts for every quad once
                flat::Plan fp;
                const std::string err = flat::plan(desc, fp);
                if (verify) check(err.empty(), "flat::plan: " + err);
                if (err.empty() && fp.flat) {
                    size_t quads_in_boxes = 0;
                    for (size_t b = 0; b < fp.boxes.size() / 32; ++b)
                        for (int f = 18; f < 24; ++f) {
                            uint32_t ref;
                            std::memcpy(&ref, &fp.boxes[32 * b + f], 4);
                            quads_in_boxes += ref != RT_REF_NONE;
                        }
                    size_t list_quads = 0;
                    const rt_node &root = desc->nodes[desc->root & RT_REF_INDEX_MASK];
                    for (uint32_t k = 0; k < root.count; ++k) list_quads += (desc->refs[root.left + k] >> RT_REF_KIND_SHIFT) == RT_KIND_QUAD;
                    check(fp.n_quads + fp.rects.size() / 8 + quads_in_boxes == list_quads, "flat::plan: quads of the list not accounted for once");
                    check(fp.n_boxes_aligned + fp.n_boxes_upright <= fp.boxes.size() / 32, "flat::plan: class counts");
                    check(fp.n_rects_x + fp.n_rects_y <= fp.rects.size() / 8, "flat::plan: rectangle axis counts");
                }
            }
            for (uint32_t builder : {uint32_t(RT_BUILD_REFERENCE), uint32_t(RT_BUILD_SAH)}) {
                std::vector<rt::Node4> nodes;
                uint32_t root = 0;
                rt::accel::Stats st;
                std::string err = rt::accel::build_wide_bvh(desc, builder, nodes, root, st);
                if (verify) check(err.empty(), "build_wide_bvh: " + err);
                if (err.empty()) {
                    std::string v = rt::accel::verify_wide_bvh(desc, nodes);
                    if (verify) check(v.empty(), "verify_wide_bvh: " + v);
                    (void)rt::accel::stack_need(nodes);
                    (void)rt::accel::sa_cost(nodes);
                }
            }
        }
        if (scene_format && has_camera) {
            rt_camera rc;
            char err[512];
            (void)rth_camera_build(&cam, &rc, err, sizeof err);
        }
    } else {
        const char *msg = rth_last_error(s);
        check(msg && msg[0], "a failed load leaves a message");
    }
    rth_scene_free(s);
    return ok;
}

std::string mutate(const std::string &src) {
    std::string t = src;
    int n = 1 + int(rnd() % 4);
    static const char *tokens[] = {"[", "]", "[[", "]]", "=", "\"", ",", "\n", "{", "}", "1e999", "-0", "nan", "inf", "0x", "objects", "type",
                                   "\"Sphere\"", "\"Group\"", "\"ConstantMedium\"", "bvh = true", "light = true", "rotate_y = 1e30", "#", "''"};
    for (int i = 0; i < n && !t.empty(); ++i) {
        size_t pos = rnd() % t.size();
        switch (rnd() % 7) {
        case 0: t.erase(pos, 1 + rnd() % 8); break;
        case 1: t.insert(pos, tokens[rnd() % (sizeof tokens / sizeof *tokens)]); break;
        case 2: t[pos] = char(rnd() % 256); break;
        case 3: t.resize(pos); break;
        case 4: { // duplicate a line
            size_t a = t.rfind('\n', pos), b = t.find('\n', pos);
            a = a == std::string::npos ? 0 : a;
            b = b == std::string::npos ? t.size() : b;
            t.insert(b, t.substr(a, b - a));
            break;
        }
        case 5: { // swap two bytes
            size_t q = rnd() % t.size();
            std::swap(t[pos], t[q]);
            break;
        }
        default: { // replace a number's first digit
            size_t q = t.find_first_of("0123456789", pos);
            if (q != std::string::npos) t[q] = "0987654321"[rnd() % 10];
        }
        }
    }
    return t;
}

} // namespace

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: host_sanitize <tmp dir> <cornell_box.toml> [scene files...]\n");
        return 2;
    }
    const std::string tmp = argv[1];
    // 1. the shipped files
    const std::string ref_text = slurp(argv[2]);
    check(load_and_build(ref_text, ".", false, true), "reference-format file loads");
    std::vector<std::pair<std::string, std::string>> scenes; // text, base dir
    for (int i = 3; i < argc; ++i) {
        scenes.emplace_back(slurp(argv[i]), dir_of(argv[i]));
        size_t n = 0;
        check(load_and_build(scenes.back().first, scenes.back().second, true, true, &n), std::string("scene file loads: ") + argv[i]);
        std::printf("%s: %zu primitives, wide BVH verified for both builders\n", argv[i], n);
    }
    // 2. damaged TOML
    int loaded = 0, refused = 0;
    for (int it = 0; it < 1500; ++it) (load_and_build(mutate(ref_text), ".", false, false) ? loaded : refused)++;
    for (const auto &sc : scenes) {
        if (sc.first.size() > 200000) continue; // the small ones are enough for the parser
        for (int it = 0; it < 600; ++it) (load_and_build(mutate(sc.first), sc.second, true, false) ? loaded : refused)++;
    }
    std::printf("damaged TOML: %d still loaded, %d refused with a message\n", loaded, refused);
    check(refused > 100 && loaded > 100, "fuzzing reached both outcomes");
    // 3. PNG
    {
        const uint32_t w = 37, h = 11;
        std::vector<uint8_t> px(size_t(w) * h * 3);
        for (auto &b : px) b = uint8_t(rnd());
        const std::string path = tmp + "/sanitize.png";
        check(rth_write_png_rgb8(path.c_str(), w, h, px.data()) == 0, "png write");
        uint32_t rw = 0, rh = 0;
        check(rth_read_png_rgb8(path.c_str(), &rw, &rh, nullptr, 0) == 0 && rw == w && rh == h, "png size query");
        std::vector<uint8_t> back(px.size());
        check(rth_read_png_rgb8(path.c_str(), &rw, &rh, back.data(), back.size()) == 0 && back == px, "png round trip");
        check(rth_read_png_rgb8(path.c_str(), &rw, &rh, back.data(), back.size() - 1) != 0, "png reader respects the capacity");
        const std::string file = slurp(path);
        int opened = 0;
        for (int it = 0; it < 400; ++it) {
            std::string bad = file;
            int n = 1 + int(rnd() % 3);
            for (int k = 0; k < n; ++k) {
                size_t pos = rnd() % bad.size();
                if (rnd() % 4 == 0) bad.resize(pos);
                else bad[pos] = char(rnd());
                if (bad.empty()) bad = "x";
            }
            const std::string p2 = tmp + "/sanitize_bad.png";
            std::ofstream(p2, std::ios::binary).write(bad.data(), std::streamsize(bad.size()));
            if (rth_read_png_rgb8(p2.c_str(), &rw, &rh, back.data(), back.size()) == 0) ++opened;
        }
        std::printf("damaged PNG: %d of 400 still decoded\n", opened);
    }
    // 4. camera builder
    {
        const double bad[] = {0.0, -1.0, 1e308, -1e308, NAN, INFINITY, 1e-320, 360.0};
        for (double v : bad)
            for (int field = 0; field < 6; ++field) {
                rth_camera_builder b;
                rth_camera_defaults(&b);
                b.image_width = 16, b.image_height = 9;
                switch (field) {
                case 0: b.vfov_deg = v; break;
                case 1: b.defocus_angle_deg = v; break;
                case 2: b.focal_length = v; break;
                case 3: b.center[0] = b.target[0] = v; break; // eye on the target
                case 4: b.vup[0] = b.vup[1] = b.vup[2] = v; break;
                default: b.samples_per_px = uint32_t(v > 0 && v < 4e9 ? v : 0);
                }
                rt_camera rc;
                char err[512];
                (void)rth_camera_build(&b, &rc, err, sizeof err);
                (void)rth_camera_with_aspect_ratio(&b, 16, v);
            }
    }
    if (failures) return 1;
    std::printf("host layer: clean\n");
    return 0;
}
