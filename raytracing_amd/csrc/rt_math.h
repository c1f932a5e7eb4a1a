// rt_math.h -- fp32 vector helpers, Philox4x32-10 and the per-path random stream (device side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

#define RT_PI 3.14159265358979323846f
#define RT_INV_PI 0.31830988618379067154f
#define RT_INF __builtin_huge_valf()

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 mk(const float4 &v) { return V3{v.x, v.y, v.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
// reference dot/cross are mul_add chains (vec.rs:122-124, 163-170); same shape in fp32
__device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ V3 fma3(V3 a, float s, V3 b) { // a*s + b
    return {fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)};
}
__device__ __forceinline__ float len2(V3 a) { return dot(a, a); }
// single-instruction reciprocal, reciprocal square root and square root (1 ulp)
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ V3 unit(V3 a) { return a * rsq(len2(a)); }

// ------------------------------------------------------------------ Philox4x32-10
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                      uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
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

__device__ __forceinline__ float u24(uint32_t u) { return float(u >> 8) * (1.0f / 16777216.0f); }

// Sequential uniforms for one (pixel, stratum, bounce, stream): counter = (pixel, stratum,
// bounce<<8 | stream, block); four 24-bit uniforms per Philox block, consumed in order.
// begin() computes block 0 at once: callers start the stream where every lane is still active
// and then branch, so the ten rounds run once at full occupancy instead of once per branch.
struct Rng {
    uint32_t k0, k1, pixel, sample, c2, block;
    uint32_t b[4];
    uint32_t pos;
    __device__ __forceinline__ void begin(uint32_t key0, uint32_t key1, uint32_t px, uint32_t smp,
                                          uint32_t stream, uint32_t bounce) {
        k0 = key0, k1 = key1, pixel = px, sample = smp;
        c2 = (bounce << 8) | stream;
        philox4x32_10(pixel, sample, c2, 0u, k0, k1, b);
        block = 1;
        pos = 0;
    }
    __device__ __forceinline__ float next() {
        if (pos == 4) {
            philox4x32_10(pixel, sample, c2, block, k0, k1, b);
            ++block;
            pos = 0;
        }
        uint32_t v = pos == 0 ? b[0] : (pos == 1 ? b[1] : (pos == 2 ? b[2] : b[3]));
        ++pos;
        return u24(v);
    }
    __device__ __forceinline__ float pm1() { return 2.0f * next() - 1.0f; }
};

// Philox2x32-10 (Random123): half the multiplies of the four-word generator, for streams that
// need one number per draw.
__host__ __device__ __forceinline__ void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, uint32_t out[2]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) key += 0x9E3779B9u;
        const uint64_t p = uint64_t(0xD256D193u) * c0;
        const uint32_t n0 = uint32_t(p >> 32) ^ key ^ c1;
        c1 = uint32_t(p);
        c0 = n0;
    }
    out[0] = c0, out[1] = c1;
}
// Key of a medium's stream: the render seed, the bounce and the medium's primitive id folded into
// one word (the counter holds pixel and stratum, so two samples never share a counter).
__host__ __device__ __forceinline__ uint32_t medium_key(uint32_t k0, uint32_t k1, uint32_t bounce, uint32_t medium_id) {
    return k0 + 0x9E3779B9u * (k1 ^ (bounce * 0x85EBCA6Bu) ^ (medium_id * 0xC2B2AE35u) ^ 0x27D4EB2Fu);
}
// One draw per (segment, medium), independent of the order a traversal meets the media in.
__device__ __forceinline__ float medium_uniform(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample,
                                                uint32_t bounce, uint32_t medium_id) {
    uint32_t o[2];
    philox2x32_10(pixel, sample, medium_key(k0, k1, bounce, medium_id), o);
    return u24(o[0]);
}

// vec.rs:211-223 (rejection in [-1,1]^3; the 1e-160 floor underflows to 0 in fp32)
__device__ __forceinline__ V3 random_in_unit_sphere(Rng &r) {
    for (;;) {
        V3 p;
        p.x = r.pm1();
        p.y = r.pm1();
        p.z = r.pm1();
        float l2 = p.x * p.x + p.y * p.y + p.z * p.z;
        if (0.0f < l2 && l2 < 1.0f) return p;
    }
}

// onb.rs:13-45
struct ONB {
    V3 u, v, w;
    // unit_already: n has length 1 to fp32 rounding (a stored normal); onb.rs normalises regardless
    __device__ __forceinline__ explicit ONB(V3 n, bool unit_already = false) {
        w = unit_already ? n : unit(n);
        V3 a = fabsf(w.x) > 0.9f ? V3{0.f, 1.f, 0.f} : V3{1.f, 0.f, 0.f};
        v = unit(cross(w, a));
        u = cross(w, v);
    }
    __device__ __forceinline__ V3 transform(V3 p) const { return (p.x * u) + (p.y * v) + (p.z * w); }
};

} // namespace rt
