// image_io.cpp -- ImageWriter counterparts (export.rs) and the PNG reader ImageTexture::load
// needs (texture.rs:95-122).  Uses zlib for deflate/inflate; no libpng.
#include "raytracing.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>

namespace raytracing {
namespace {

void put32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)), v.push_back(uint8_t(x >> 16)), v.push_back(uint8_t(x >> 8)), v.push_back(uint8_t(x));
}
uint32_t get32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

void chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data) {
    put32(out, uint32_t(data.size()));
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    uint32_t crc = uint32_t(crc32(0L, out.data() + start, uInt(out.size() - start)));
    put32(out, crc);
}

std::vector<uint8_t> quantize(uint32_t w, uint32_t h, const float *rgb, bool gamma) {
    std::vector<uint8_t> px(size_t(w) * h * 3);
    for (size_t i = 0; i < size_t(w) * h; ++i) {
        Color c{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        if (gamma) c = c.as_gamma_corrected();
        auto q = c.as_rgb_ints();
        px[3 * i] = q[0], px[3 * i + 1] = q[1], px[3 * i + 2] = q[2];
    }
    return px;
}

uint8_t paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return uint8_t((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

} // namespace

// PngWriter::write (export.rs:77-86): linear values, 8-bit RGB
void write_png(const std::string &path, uint32_t w, uint32_t h, const float *rgb) {
    write_png_rgb8(path, w, h, quantize(w, h, rgb, false).data());
}

void write_png_rgb8(const std::string &path, uint32_t w, uint32_t h, const uint8_t *rgb8) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t(w) * 3 + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0); // filter: none
        raw.insert(raw.end(), rgb8 + size_t(y) * w * 3, rgb8 + size_t(y + 1) * w * 3);
    }
    uLongf clen = compressBound(uLong(raw.size()));
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), uLong(raw.size()), 6) != Z_OK) throw Error("write_png: deflate failed");
    comp.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put32(ihdr, w), put32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0}); // depth 8, colour type RGB
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", comp);
    chunk(out, "IEND", {});
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error("write_png: cannot open " + path);
    size_t n = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    if (n != out.size()) throw Error("write_png: short write to " + path);
}

// PpmWriter (export.rs:28-41): P3, gamma 2
void write_ppm(const std::string &path, uint32_t w, uint32_t h, const float *rgb) {
    write_ppm_rgb8(path, w, h, quantize(w, h, rgb, true).data());
}

void write_ppm_rgb8(const std::string &path, uint32_t w, uint32_t h, const uint8_t *px) {
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) throw Error("write_ppm: cannot open " + path);
    std::fprintf(f, "P3\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < size_t(w) * h; ++i) std::fprintf(f, "%u %u %u\n", px[3 * i], px[3 * i + 1], px[3 * i + 2]);
    std::fclose(f);
}

// 8-bit RGB, non-interlaced, non-animated PNG only -- what ImageTexture::load accepts
void read_png_rgb8(const std::string &path, std::vector<uint8_t> &rgb, uint32_t &w, uint32_t &h) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw Error("Failed to find file " + path);
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw Error("read_png: not a PNG file: " + path);
    std::vector<uint8_t> idat;
    bool have_ihdr = false;
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        uint32_t len = get32(&file[pos]);
        if (pos + 12 + size_t(len) > file.size()) throw Error("read_png: truncated chunk");
        const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) throw Error("read_png: bad IHDR");
            w = get32(data), h = get32(data + 4);
            if (data[8] != 8 || data[9] != 2) throw Error("Must be 8-bit PNG."); // texture.rs:103-106
            if (data[12] != 0) throw Error("read_png: interlaced PNG not supported");
            if (w == 0 || h == 0 || uint64_t(w) * h > (1ull << 28)) throw Error("read_png: bad dimensions");
            have_ihdr = true;
        } else if (!std::memcmp(type, "acTL", 4)) {
            throw Error("Cannot accept APNGs."); // texture.rs:99-102
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + size_t(len);
    }
    if (!have_ihdr) throw Error("read_png: missing IHDR");
    size_t stride = size_t(w) * 3;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rlen = uLongf(raw.size());
    if (uncompress(raw.data(), &rlen, idat.data(), uLong(idat.size())) != Z_OK || rlen != raw.size())
        throw Error("read_png: inflate failed");
    rgb.assign(stride * h, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *src = &raw[(stride + 1) * y + 1];
        uint8_t *dst = &rgb[stride * y];
        const uint8_t *up = y ? &rgb[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= 3 ? dst[x - 3] : 0, b = up ? up[x] : 0, c = (up && x >= 3) ? up[x - 3] : 0;
            uint8_t v = src[x];
            switch (ft) {
            case 0: break;
            case 1: v = uint8_t(v + a); break;
            case 2: v = uint8_t(v + b); break;
            case 3: v = uint8_t(v + ((a + b) >> 1)); break;
            case 4: v = uint8_t(v + paeth(a, b, c)); break;
            default: throw Error("read_png: bad filter type");
            }
            dst[x] = v;
        }
    }
}

} // namespace raytracing
