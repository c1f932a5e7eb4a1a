// rt_device.hip -- device half of the C ABI (rt_* in include/rt_amd.h): scene upload,
// the wavefront render loop, film handling and the parity hooks.
#include "rt_kernels.h"
#include "rt_accel.h"
#include "rt_flat.h"

#include <rccl/rccl.h> // types only: the library is loaded when a group is formed (rt_comm_*)

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

using namespace rt;

namespace {

thread_local std::string g_create_error;

struct EventPair {
    hipEvent_t a, b;
    int cls;
};
// Event pairs a handle keeps for timing launches: drained at every poll point of the ray-queue loop,
// so a long render reuses a few dozen; the cap is for a caller that times everything and never polls.
constexpr size_t EV_POOL_MAX = 4096;

} // namespace

struct rt_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // scene
    std::vector<void *> scene_allocs;
    DScene sc{};           // host copy of the descriptor the kernels read from d_sc
    DScene *d_sc = nullptr;
    DCamera *d_cam = nullptr;
    bool have_scene = false;
    bool flat = false;
    int accel = ACCEL_BVH2;                 // structure the kernels search for this scene
    uint32_t build = RT_BUILD_AUTO, width = 4; // rt_set_accel: applies to the next upload
    uint32_t stack4 = 0;                    // LDS stack entries per lane of the wide traversal
    uint32_t lds_nodes4 = 0;                // wide nodes staged in LDS behind the stack
    size_t node4_bytes() const { return accel == ACCEL_BVH4Q ? sizeof(Node4Q) : sizeof(Node4); }
    size_t dyn_lds4() const { return size_t(stack4) * BLOCK * sizeof(uint32_t) + size_t(lds_nodes4) * node4_bytes(); }
    enum { BOXES_AUTO = 0, BOXES_FP32 = 1, BOXES_BYTE = 2 };
    int boxes = BOXES_AUTO;                 // rt_set_accel: RT_BUILD_FP32_BOXES / RT_BUILD_BYTE_BOXES
    bool box_leaves = true;                 // rt_set_accel: RT_BUILD_NO_BOX_LEAVES clears it
    bool post_prims = true;                 // rt_set_accel: RT_BUILD_NO_POST_TESTS clears it
    uint32_t n_box_leaves = 0;              // leaves of kind RT_KIND_BOX in the tree searched
    accel::Stats accel_stats{};
    uint32_t has_media = 0, n_prims = 0, alone_spheres = 0, n_quads = 0, n_spheres = 0, n_materials = 0;
    bool has_specular = false;
    bool has_motion = false; // some sphere moves: a ray's time matters
    uint32_t last_variant = 0;    // RT_VARIANT_* of the persistent kernel the last rt_render took
    bool plain_materials = false; // no Isotropic material, every texture a colour folded into its material's record
    uint32_t n_textures = 0;
    std::vector<uint32_t> prim_to_ref; // user-visible primitive id -> ref (parity hook)

    // queue
    Queue q{};
    size_t q_slots = 0;
    Ctrl *d_ctrl = nullptr;
    Ctrl *h_ctrl = nullptr; // pinned, two slots

    // film
    float *film = nullptr;
    bool film_owned = false;
    uint32_t film_w = 0, film_h = 0;

    // device staging of the boundary calls that move host arrays (film download / upload /
    // quantize, the parity hooks): one buffer kept on the handle, grown when a call needs more
    void *stage = nullptr;
    size_t stage_bytes = 0;
    // global-memory part of the traversal stacks of k_paths_pool (entries beyond POOL_STACK)
    uint32_t *spill = nullptr;
    size_t spill_bytes = 0;
    // this device's copy of `scratch` (below): rt_film_quantize_noalloc
    uint32_t *noalloc_scratch = nullptr;

    // the group this handle reduces its film with (rt_comm_*): an RCCL communicator
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 0;
    unsigned long long comm_group = 0; // which rt_comm_init_all made it (0: rt_comm_init_rank)

    // stats
    rt_stats stats{};
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_poll[2] = {nullptr, nullptr};
};

// Scratch of rt_film_quantize_noalloc, the entry point that allocates nothing: the packed 8-bit
// image on its way to the host.  Every GPU the library runs on holds its own copy of this array
// from the moment the code object is loaded until the process ends.  The name is a host-side
// stand-in, not a device address: host code uses the pointer hipGetSymbolAddress returns for the
// current device (scratch_for_device), never the name.
constexpr uint32_t NOALLOC_MAX_PIXELS = 3840u * 2160u; // one UHD frame: 24.9 MB of bytes
static __device__ __align__(256) uint32_t scratch[(NOALLOC_MAX_PIXELS + 3u) / 4u * 3u];

namespace {

int fail(rt_handle *h, const std::string &msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return -1;
}

#define RT_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(h, std::string(#call) + ": " + hipGetErrorString(e_));               \
    } while (0)

// rt_render_params as the caller's header defines it -> as this library does.  Fields are only ever
// appended: a shorter struct comes from an older header and the fields it lacks are 0 (the defaults);
// a longer one holds fields this library cannot honour and is refused.
int read_params(rt_handle *h, const rt_render_params *in, rt_render_params *out) {
    if (!in) return fail(h, "null render params");
    constexpr uint32_t oldest = uint32_t(offsetof(rt_render_params, seed) + sizeof(uint64_t));
    if (in->struct_size < oldest)
        return fail(h, "rt_render_params.struct_size is " + std::to_string(in->struct_size) + ": set it to sizeof(rt_render_params) (header ABI " +
                           std::to_string(RT_ABI_VERSION) + "; a binding built against an older header must be rebuilt)");
    if (in->struct_size > sizeof(rt_render_params))
        return fail(h, "rt_render_params comes from a newer header (" + std::to_string(in->struct_size) + " bytes) than this library (" +
                           std::to_string(sizeof(rt_render_params)) + " bytes, ABI " + std::to_string(RT_ABI_VERSION) + ")");
    *out = rt_render_params{};
    std::memcpy(out, in, in->struct_size);
    out->struct_size = uint32_t(sizeof(rt_render_params));
    return 0;
}
// An output struct that starts with struct_size: the caller says how many bytes it has room for,
// the library fills the part both know and says how many it filled.  Returns "" or what is wrong.
template <typename T> std::string write_sized(const T &full, T *out, const char *name) {
    const uint32_t room = out->struct_size;
    if (room < 2 * sizeof(uint32_t))
        return std::string(name) + ".struct_size is " + std::to_string(room) + ": set it to sizeof(" + name + ") before the call (header ABI " +
               std::to_string(RT_ABI_VERSION) + ")";
    T tmp = full;
    tmp.struct_size = uint32_t(std::min<size_t>(room, sizeof(T)));
    std::memcpy(out, &tmp, tmp.struct_size);
    return "";
}

// The address of `scratch` on a device, looked up once per device with that device current.
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
uint32_t *g_scratch_ptr[MAX_DEVICES] = {};
// all handles of a device stage through that device's one array: one call at a time per device
std::mutex g_scratch_in_use[MAX_DEVICES];
int scratch_for_device(rt_handle *h, int device, uint32_t **out) {
    if (device < 0 || device >= MAX_DEVICES) return fail(h, "device id beyond the scratch table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        RT_HIP(hipSetDevice(device));
        RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(h, "hipGetSymbolAddress(scratch) returned a null pointer");
        g_scratch_ptr[device] = static_cast<uint32_t *>(p);
    }
    *out = g_scratch_ptr[device];
    return 0;
}

// Device staging kept on the handle: grows to the largest request seen, freed by rt_destroy.
int ensure_stage(rt_handle *h, size_t bytes, void **out) {
    if (bytes > h->stage_bytes) {
        RT_HIP(hipStreamSynchronize(h->stream)); // nothing in flight may still use the old buffer
        if (h->stage) (void)hipFree(h->stage);
        h->stage = nullptr, h->stage_bytes = 0;
        const size_t want = (bytes + 0xFFFFu) & ~size_t(0xFFFFu);
        RT_HIP(hipMalloc(&h->stage, want));
        h->stage_bytes = want;
    }
    *out = h->stage;
    return 0;
}

template <typename T> int upload(rt_handle *h, const T *src, size_t n, const T **dst) {
    *dst = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    void *p = nullptr;
    RT_HIP(hipMalloc(&p, bytes));
    h->scene_allocs.push_back(p);
    if (n) RT_HIP(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
    else RT_HIP(hipMemsetAsync(p, 0, bytes, h->stream));
    *dst = static_cast<const T *>(p);
    return 0;
}

void free_scene(rt_handle *h) {
    for (void *p : h->scene_allocs) (void)hipFree(p);
    h->scene_allocs.clear();
    h->have_scene = false;
}

void free_queue(rt_handle *h) {
    void *ptrs[] = {h->q.ra, h->q.rb, h->q.sa, h->q.sb, h->q.hit, h->q.count, h->q.gen_base, h->q.gen_n};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    h->q = Queue{};
    h->q_slots = 0;
}

bool ref_ok(const rt_scene_desc *s, uint32_t r, bool allow_node, bool allow_medium) {
    uint32_t kind = r >> RT_REF_KIND_SHIFT, idx = r & RT_REF_INDEX_MASK;
    switch (kind) {
    case RT_KIND_NODE: return allow_node && idx < s->n_nodes;
    case RT_KIND_QUAD: return idx < s->n_quads;
    case RT_KIND_SPHERE: return idx < s->n_spheres;
    case RT_KIND_MEDIUM: return allow_medium && idx < s->n_media;
    default: return false;
    }
}

// Everything the kernels index with is checked here, once, on the host.
int validate_scene(rt_handle *h, const rt_scene_desc *s) {
    if (!s) return fail(h, "rt_upload_scene: null scene");
    if ((s->n_nodes && !s->nodes) || (s->n_refs && !s->refs) || (s->n_quads && !s->quads) ||
        (s->n_spheres && !s->spheres) || (s->n_media && !s->media) || (s->n_materials && !s->materials) ||
        (s->n_textures && !s->textures) || (s->n_lights && !s->light_refs) || (s->n_texel_bytes && !s->texels) ||
        (s->n_noise && (!s->noise_vec || !s->noise_perm)))
        return fail(h, "rt_upload_scene: null array with non-zero count");
    if (s->n_materials == 0) return fail(h, "rt_upload_scene: scene has no materials");
    if (!ref_ok(s, s->root, true, true)) return fail(h, "rt_upload_scene: root reference out of range");
    for (uint32_t i = 0; i < s->n_nodes; ++i) {
        const rt_node &n = s->nodes[i];
        if (n.right == RT_NODE_LEAF) {
            if (uint64_t(n.left) + n.count > s->n_refs) return fail(h, "rt_upload_scene: leaf range outside refs");
        } else {
            if (!ref_ok(s, n.left, true, true) || !ref_ok(s, n.right, true, true))
                return fail(h, "rt_upload_scene: node child out of range");
            // a child index must point forward, or traversal could loop
            for (uint32_t c : {n.left, n.right})
                if ((c >> RT_REF_KIND_SHIFT) == RT_KIND_NODE && (c & RT_REF_INDEX_MASK) <= i)
                    return fail(h, "rt_upload_scene: node child does not point forward");
        }
    }
    for (uint32_t i = 0; i < s->n_refs; ++i) {
        if (!ref_ok(s, s->refs[i], true, true)) return fail(h, "rt_upload_scene: ref out of range");
    }
    // leaf entries that are nodes must point forward as well
    for (uint32_t i = 0; i < s->n_nodes; ++i) {
        const rt_node &n = s->nodes[i];
        if (n.right != RT_NODE_LEAF) continue;
        for (uint32_t k = 0; k < n.count; ++k) {
            uint32_t c = s->refs[n.left + k];
            if ((c >> RT_REF_KIND_SHIFT) == RT_KIND_NODE && (c & RT_REF_INDEX_MASK) <= i)
                return fail(h, "rt_upload_scene: list entry does not point forward");
        }
    }
    for (uint32_t i = 0; i < s->n_quads; ++i) {
        if (s->quads[i].mat >= s->n_materials) return fail(h, "rt_upload_scene: quad material out of range");
        if (s->quads[i].shape > RT_SHAPE_DISC) return fail(h, "rt_upload_scene: bad quad shape");
    }
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        if (s->spheres[i].mat >= s->n_materials) return fail(h, "rt_upload_scene: sphere material out of range");
        if (!(s->spheres[i].radius > 0.f)) return fail(h, "rt_upload_scene: sphere radius must be > 0");
    }
    for (uint32_t i = 0; i < s->n_media; ++i) {
        const rt_medium &m = s->media[i];
        if (m.mat >= s->n_materials) return fail(h, "rt_upload_scene: medium material out of range");
        if (uint64_t(m.first) + m.count > s->n_refs || m.count == 0)
            return fail(h, "rt_upload_scene: medium boundary outside refs");
        for (uint32_t k = 0; k < m.count; ++k)
            if (!ref_ok(s, s->refs[m.first + k], false, false))
                return fail(h, "rt_upload_scene: medium boundary must be quads/spheres");
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const rt_material &m = s->materials[i];
        if (m.type > RT_MAT_ISOTROPIC) return fail(h, "rt_upload_scene: bad material type");
        bool textured = m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_LIGHT || m.type == RT_MAT_ISOTROPIC;
        if (textured && m.tex >= s->n_textures) return fail(h, "rt_upload_scene: material texture out of range");
    }
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        const rt_texture &t = s->textures[i];
        if (t.type > RT_TEX_NOISE) return fail(h, "rt_upload_scene: bad texture type");
        if (t.type == RT_TEX_CHECKER) {
            // children must have smaller ids (they exist before their parent), so chains end
            if (t.a >= i || t.b >= i) return fail(h, "rt_upload_scene: checker child must precede it");
        }
        if (t.type == RT_TEX_IMAGE) {
            if (t.a == 0 || t.b == 0 || uint64_t(t.offset) + uint64_t(t.a) * t.b * 3 > s->n_texel_bytes)
                return fail(h, "rt_upload_scene: image texture outside texel buffer");
        }
        if (t.type == RT_TEX_NOISE && t.offset >= s->n_noise)
            return fail(h, "rt_upload_scene: noise table out of range");
    }
    if (s->n_noise)
        for (size_t i = 0; i < size_t(s->n_noise) * 768; ++i)
            if (s->noise_perm[i] < 0 || s->noise_perm[i] > 255)
                return fail(h, "rt_upload_scene: noise permutation entry outside 0..255");
    if ((s->lights_kind == RT_LIGHTS_NONE) != (s->n_lights == 0) || s->lights_kind > RT_LIGHTS_LIST)
        return fail(h, "rt_upload_scene: lights_kind does not match n_lights");
    if (s->lights_kind == RT_LIGHTS_SINGLE && s->n_lights != 1)
        return fail(h, "rt_upload_scene: single light needs n_lights == 1");
    for (uint32_t i = 0; i < s->n_lights; ++i)
    {
        if (!ref_ok(s, s->light_refs[i], false, false)) return fail(h, "rt_upload_scene: light must be a quad or sphere");
        const uint32_t lr = s->light_refs[i];
        // only Parallelogram implements pdf_value / random (hittable.rs:377-395)
        if ((lr >> RT_REF_KIND_SHIFT) == RT_KIND_QUAD && s->quads[lr & RT_REF_INDEX_MASK].shape != RT_SHAPE_PARALLELOGRAM)
            return fail(h, "rt_upload_scene: a quad light must be a parallelogram");
        // (the light records tell a sphere by its radius)
        if ((lr >> RT_REF_KIND_SHIFT) == RT_KIND_SPHERE && !(s->spheres[lr & RT_REF_INDEX_MASK].radius > 0.0f))
            return fail(h, "rt_upload_scene: a sphere light needs a positive radius");
    }
    return 0;
}

// Stack entries the binary traversal (traverse) can need for this descriptor, whatever the caller
// says in max_stack: an interior node pushes its far child and goes on with the near one; a list
// pushes every entry that is a node and then pops them from the top.  Children point forward
// (validated), so one backward pass does it.
uint32_t binary_stack_need(const rt_scene_desc *s) {
    std::vector<uint32_t> need(s->n_nodes, 0);
    auto of = [&](uint32_t ref) { return (ref >> RT_REF_KIND_SHIFT) == RT_KIND_NODE ? need[ref & RT_REF_INDEX_MASK] : 0u; };
    for (uint32_t i = s->n_nodes; i-- > 0;) {
        const rt_node &n = s->nodes[i];
        if (n.right != RT_NODE_LEAF) {
            need[i] = 1u + std::max(of(n.left), of(n.right));
        } else {
            uint32_t pushed = 0, worst = 0;
            for (uint32_t k = 0; k < n.count; ++k) {
                const uint32_t r = s->refs[n.left + k];
                if ((r >> RT_REF_KIND_SHIFT) != RT_KIND_NODE) continue;
                worst = std::max(worst, pushed + need[r & RT_REF_INDEX_MASK]); // entries below it are still on the stack
                ++pushed;
            }
            need[i] = std::max(worst, pushed);
        }
    }
    return of(s->root);
}

int make_camera(rt_handle *h, const rt_camera *c, const rt_render_params *p, DCamera *out, uint32_t *se, bool timeless = false) {
    if (!c || !p) return fail(h, "null camera or params");
    if (c->image_width < 1 || c->image_height < 1) return fail(h, "camera: empty image");
    if (uint64_t(c->image_width) * c->image_height > (1ull << 28)) return fail(h, "camera: image too large");
    if (c->sqrt_spp < 1 || c->sqrt_spp > 4096) return fail(h, "camera: sqrt_spp must be in 1..4096");
    if (c->max_depth < 1 || c->max_depth > 0xFFFFFFu) return fail(h, "camera: max_depth must be at least 1");
    uint32_t spp = c->sqrt_spp * c->sqrt_spp;
    uint32_t sb = p->stratum_begin, e = p->stratum_end ? p->stratum_end : spp;
    if (e > spp || sb > e) return fail(h, "params: stratum range outside 0..sqrt_spp^2");
    DCamera d{};
    auto f3 = [](const double *v) { return V3{float(v[0]), float(v[1]), float(v[2])}; };
    double rel[3];
    for (int i = 0; i < 3; ++i) rel[i] = c->pixel_00[i] - c->center[i]; // f64 difference, then fp32
    d.p00_rel = f3(rel);
    d.du = f3(c->pxdelta_u);
    d.dv = f3(c->pxdelta_v);
    d.center = f3(c->center);
    d.disk_u = f3(c->defocus_disk_u);
    d.disk_v = f3(c->defocus_disk_v);
    d.bg = f3(c->background);
    d.W = c->image_width;
    d.H = c->image_height;
    d.npix = d.W * d.H;
    d.sqrt_spp = c->sqrt_spp;
    d.inv_sqrt = float(1.0 / double(c->sqrt_spp));
    d.inv_w = float(1.0 / double(c->image_width));
    d.stratum_begin = sb;
    d.max_depth = c->max_depth;
    d.defocus = !(c->defocus_angle <= 0.0); // camera.rs:422
    d.sky = c->background_sky != 0;
    d.aa_disc = c->antialiasing == RT_AA_DISC;
    d.timeless = timeless ? 1u : 0u; // the time of a ray is not drawn (camera_ray)
    *out = d;
    *se = e;
    // the kernels read the camera from device memory
    RT_HIP(hipMemcpyAsync(h->d_cam, out, sizeof(DCamera), hipMemcpyHostToDevice, h->stream));
    RT_HIP(hipStreamSynchronize(h->stream)); // *out is the caller's stack
    return 0;
}

int ensure_queue(rt_handle *h, uint32_t n_chunks, uint32_t cap) {
    size_t slots = size_t(n_chunks) * cap;
    if (h->q_slots >= slots && h->q.n_chunks >= n_chunks && h->q.cap == cap) {
        h->q.n_chunks = n_chunks;
        return 0;
    }
    free_queue(h);
    RT_HIP(hipMalloc((void **)&h->q.ra, slots * sizeof(float4)));
    RT_HIP(hipMalloc((void **)&h->q.rb, slots * sizeof(float4)));
    RT_HIP(hipMalloc((void **)&h->q.sa, slots * sizeof(float4)));
    RT_HIP(hipMalloc((void **)&h->q.sb, slots * sizeof(uint2)));
    RT_HIP(hipMalloc((void **)&h->q.hit, slots * sizeof(float4)));
    RT_HIP(hipMalloc((void **)&h->q.count, n_chunks * sizeof(uint32_t)));
    RT_HIP(hipMalloc((void **)&h->q.gen_base, n_chunks * sizeof(unsigned long long)));
    RT_HIP(hipMalloc((void **)&h->q.gen_n, n_chunks * sizeof(uint32_t)));
    h->q.cap = cap;
    h->q.n_chunks = n_chunks;
    h->q_slots = slots;
    return 0;
}

int ensure_film(rt_handle *h, uint32_t w, uint32_t hgt) {
    if (h->film && h->film_w == w && h->film_h == hgt) return 0;
    if (h->film && !h->film_owned)
        return fail(h, "film: bound film has different dimensions than the camera");
    if (h->film) (void)hipFree(h->film);
    h->film = nullptr;
    size_t n = size_t(w) * hgt * 3;
    RT_HIP(hipMalloc((void **)&h->film, n * sizeof(float)));
    RT_HIP(hipMemsetAsync(h->film, 0, n * sizeof(float), h->stream));
    h->film_owned = true;
    h->film_w = w;
    h->film_h = hgt;
    return 0;
}


// timed launch helper
struct Timer {
    rt_handle *h;
    bool on;
    int begin(int cls) {
        if (!on) return -1;
        if (h->ev_used == h->ev_pool.size()) {
            if (h->ev_pool.size() >= EV_POOL_MAX) return -1; // this launch goes untimed (and uncounted)
            EventPair p{};
            if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return -1;
            h->ev_pool.push_back(p);
        }
        int i = int(h->ev_used++);
        h->ev_pool[i].cls = cls;
        h->stats.timed_launches[cls]++;
        (void)hipEventRecord(h->ev_pool[i].a, h->stream);
        return i;
    }
    void end(int i) {
        if (i >= 0) (void)hipEventRecord(h->ev_pool[i].b, h->stream);
    }
};

// Adds up the first `upto` event pairs (their launches are known to be complete) and hands them
// back to the pool; pairs recorded later move to the front.
void drain_timers(rt_handle *h, size_t upto) {
    upto = std::min(upto, h->ev_used);
    for (size_t i = 0; i < upto; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b) == hipSuccess)
            h->stats.kernel_ms[h->ev_pool[i].cls] += ms;
    }
    std::rotate(h->ev_pool.begin(), h->ev_pool.begin() + upto, h->ev_pool.begin() + h->ev_used);
    h->ev_used -= upto;
}
void drain_timers(rt_handle *h) { drain_timers(h, h->ev_used); }

// One launch of the persistent path kernel: as many blocks as the device keeps resident.
template <int MODE, int ACCEL, bool EXTRAS = true, bool PLAIN = false, bool SIMPLE = false>
int launch_paths(rt_handle *h, const DCamera &dc, const rt_render_params *p, const Film &film,
                 unsigned long long total) {
    auto kernel = k_paths<MODE, ACCEL, EXTRAS, PLAIN, SIMPLE>;
    int per_cu = 0, n_cu = 0;
    const size_t dyn_lds = accel_is_wide(ACCEL) ? h->dyn_lds4() : 0;
    RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, dyn_lds));
    RT_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
    if (per_cu < 1 || n_cu < 1) return fail(h, "rt_render: path kernel does not fit on a compute unit");
    if (p->waves_per_simd && int(p->waves_per_simd) < per_cu) per_cu = int(p->waves_per_simd); // a block is one wave per SIMD
    // most samples a wave takes from the global counter at a time (the kernel takes fewer as the
    // work runs out); 1024 keeps the counter idle: one atomic per 1024 paths
    const unsigned long long grab = 1024;
    unsigned long long blocks_needed = (total + WAVE * (BLOCK / WAVE) - 1) / (WAVE * (BLOCK / WAVE));
    unsigned long long blocks = (unsigned long long)per_cu * n_cu;
    if (blocks > blocks_needed) blocks = blocks_needed;
    uint32_t refill_min = p->refill_min ? p->refill_min : 8u;
    if (refill_min > uint32_t(WAVE)) refill_min = WAVE;
    Timer tm{h, (p->flags & RT_FLAG_TIME_KERNELS) != 0};
    int e = tm.begin(4);
    kernel<<<dim3(uint32_t(blocks)), dim3(BLOCK), dyn_lds, h->stream>>>(h->d_sc, h->d_cam, film, uint32_t(p->seed), uint32_t(p->seed >> 32),
                                                                h->d_ctrl, h->has_media,
                                                                uint32_t(grab), refill_min);
    tm.end(e);
    RT_HIP(hipGetLastError());
    h->stats.launches[4]++;
    return 0;
}

// The regrouping variant of the persistent kernel (flat worlds): same launch shape.
template <int MODE, bool TABLES, bool TIMELESS, bool BOXED = false, uint32_t LIGHTS = LIGHTS_ANY>
int launch_paths_regroup(rt_handle *h, const rt_render_params *p, const Film &film, unsigned long long total) {
    auto kernel = k_paths_regroup<MODE, TABLES, TIMELESS, BOXED, LIGHTS>;
    int per_cu = 0, n_cu = 0;
    RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, 0));
    RT_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
    if (per_cu < 1 || n_cu < 1) return fail(h, "rt_render: path kernel does not fit on a compute unit");
    if (p->waves_per_simd && int(p->waves_per_simd) < per_cu) per_cu = int(p->waves_per_simd);
    const unsigned long long per_block = (unsigned long long)RegroupPool<TIMELESS>::SLOTS * (BLOCK / WAVE);
    unsigned long long blocks_needed = (total + per_block - 1) / per_block;
    unsigned long long blocks = (unsigned long long)per_cu * n_cu;
    if (blocks > blocks_needed) blocks = blocks_needed;
    // every wave's first range of samples comes with the launch (k_paths_regroup): a quarter of an even
    // share like the ranges the kernel asks for, whole waves of 64, at most 1024; none when the frame is
    // too small for that
    const unsigned long long n_waves = blocks * (BLOCK / WAVE);
    unsigned long long first_take = std::min<unsigned long long>(total / (4 * n_waves), 1024) & ~63ull;
    if (first_take < MIN_TAKE_FLAT) first_take = MIN_TAKE_FLAT;
    if (first_take * n_waves > total) first_take = 0;
    Timer tm{h, (p->flags & RT_FLAG_TIME_KERNELS) != 0};
    int e = tm.begin(4);
    kernel<<<dim3(uint32_t(blocks)), dim3(BLOCK), 0, h->stream>>>(h->d_sc, h->d_cam, film, uint32_t(p->seed), uint32_t(p->seed >> 32),
                                                                 h->d_ctrl, h->has_media, 1024u, h->n_quads, h->n_spheres, h->n_materials,
                                                                 uint32_t(first_take));
    tm.end(e);
    RT_HIP(hipGetLastError());
    h->stats.launches[4]++;
    h->stats.regrouped = 1;
    return 0;
}

// The regrouping kernel behind the wide BVH: a pool of paths per wave, traversals resumable per lane.
constexpr uint32_t POOL_MIN_NODES = 8192; // trees from this size take it by default
template <int MODE, bool QUANT, bool EXTRAS, bool PLAIN = false>
int launch_paths_pool(rt_handle *h, const rt_render_params *p, const Film &film, unsigned long long total) {
    auto kernel = k_paths_pool<MODE, QUANT, EXTRAS, PLAIN>;
    // what is left of a quarter of the CU's LDS behind pools and stacks holds the top of the tree
    const size_t node_bytes = QUANT ? sizeof(Node4Q) : sizeof(Node4), budget = 40 * 1024;
    // A large tree is read from memory with plain loads: the few dozen nodes that fit beside the pools make
    // every node load a generic one (LDS or memory, decided per lane), which costs more than the top of a tree
    // of hundreds of thousands of nodes saves (1 M-primitive scene: +2.9 % without; small trees -3 to -5 % without).
#ifndef RT_POOL_STAGE_MAX_NODES
#define RT_POOL_STAGE_MAX_NODES 4096u
#endif
    const uint32_t lds_nodes = POOL_LDS_FIXED < budget && h->accel_stats.nodes <= RT_POOL_STAGE_MAX_NODES
                                   ? uint32_t(std::min<size_t>(h->accel_stats.nodes, (budget - POOL_LDS_FIXED) / node_bytes)) : 0u;
    const size_t dyn_lds = POOL_LDS_FIXED + lds_nodes * node_bytes;
    int per_cu = 0, n_cu = 0;
    RT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, dyn_lds));
    RT_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
    if (per_cu < 1 || n_cu < 1) return fail(h, "rt_render: path kernel does not fit on a compute unit");
    if (p->waves_per_simd && int(p->waves_per_simd) < per_cu) per_cu = int(p->waves_per_simd);
    const unsigned long long per_block = (unsigned long long)POOL_SLOTS * (BLOCK / WAVE);
    unsigned long long blocks = (unsigned long long)per_cu * n_cu;
    blocks = std::min(blocks, (total + per_block - 1) / per_block);
    // stack entries beyond the LDS part: one column per lane of the grid
    const uint32_t deep = h->accel_stats.stack_need > POOL_STACK ? h->accel_stats.stack_need - POOL_STACK : 0u;
    const size_t need = size_t(blocks) * BLOCK * deep * sizeof(uint32_t);
    if (need > h->spill_bytes) {
        RT_HIP(hipStreamSynchronize(h->stream));
        if (h->spill) (void)hipFree(h->spill);
        h->spill = nullptr, h->spill_bytes = 0;
        RT_HIP(hipMalloc((void **)&h->spill, need));
        h->spill_bytes = need;
    }
    Timer tm{h, (p->flags & RT_FLAG_TIME_KERNELS) != 0};
    int e = tm.begin(4);
    kernel<<<dim3(uint32_t(blocks)), dim3(BLOCK), dyn_lds, h->stream>>>(h->d_sc, h->d_cam, film, uint32_t(p->seed), uint32_t(p->seed >> 32),
                                                                       h->d_ctrl, h->has_media, 1024u, h->spill, lds_nodes);
    tm.end(e);
    RT_HIP(hipGetLastError());
    h->stats.launches[4]++;
    h->stats.regrouped = 1;
    return 0;
}

int render_persistent(rt_handle *h, const DCamera &dc, const rt_render_params *p, const Film &film,
                      unsigned long long total) {
    RT_HIP(hipEventRecord(h->ev_begin, h->stream));
    int rc;
    // the regrouping kernel packs stratum and bounce into one word
    const bool regroup_fits = uint64_t(dc.sqrt_spp) * dc.sqrt_spp <= REGROUP_MAX_STRATA && dc.max_depth <= REGROUP_MAX_DEPTH;
    const bool regroup_wanted = !(p->flags & RT_FLAG_NO_REGROUP);
    // box leaves in the wide tree, or primitives kept out of it and tested after the walk: the kernels
    // behind a tree are compiled with and without the code for them
    const bool extras = h->n_box_leaves != 0 || h->sc.n_post != 0;
    // ... and, for scenes with neither, once more without the code for motion, media and textures
    const bool general = (p->flags & RT_FLAG_GENERAL_KERNELS) != 0;
    uint32_t variant_flat = 0;
    const bool plain = !general && !extras && dc.timeless && !h->has_media && h->plain_materials;
    // ... and a scene with them once more for a pinhole camera and a single light
    const bool simple = !general && extras && !dc.defocus && !dc.aa_disc && h->sc.lights_kind == RT_LIGHTS_SINGLE;
    if (h->accel == ACCEL_FLAT && regroup_wanted && regroup_fits) {
        // a small scene's quads, spheres and materials are copied into LDS tables
        const bool tables = h->n_quads <= TABLE_QUADS && h->n_spheres <= TABLE_SPHERES && h->n_materials <= TABLE_MATERIALS;
        // (in a scene in which nothing moves a slot of the pool does without the ray's time: more slots)
        // (a Cornell box -- rectangles, upright boxes, spheres in plain colours, nothing moving, no medium -- takes a
        // scan without the other loops and shading without textures)
        const bool timeless = dc.timeless && !general;
        const bool boxed = timeless && tables && !h->has_media && h->plain_materials && h->sc.n_flat_pairs == 0 && h->sc.n_flat_quads == 0 && h->sc.n_flat_other == 0 &&
                           h->sc.n_flat_boxes == h->sc.n_flat_boxes_aligned + h->sc.n_flat_boxes_upright;
#define RT_REGROUP(MODE)                                                                                                  \
    (boxed  ? (h->sc.lights_kind == RT_LIGHTS_LIST ? launch_paths_regroup<MODE, true, true, true, RT_LIGHTS_LIST>(h, p, film, total) \
                                                   : launch_paths_regroup<MODE, true, true, true>(h, p, film, total)) \
     : tables ? (timeless ? launch_paths_regroup<MODE, true, true>(h, p, film, total) : launch_paths_regroup<MODE, true, false>(h, p, film, total)) \
            : (timeless ? launch_paths_regroup<MODE, false, true>(h, p, film, total) : launch_paths_regroup<MODE, false, false>(h, p, film, total)))
        rc = p->mode == RT_MODE_REF_EXACT ? RT_REGROUP(RT_MODE_REF_EXACT) : RT_REGROUP(RT_MODE_BOOK3);
#undef RT_REGROUP
        variant_flat = (timeless ? RT_VARIANT_TIMELESS : 0u) | (boxed ? RT_VARIANT_BOXED : 0u);
    } else if (accel_is_wide(h->accel) && regroup_fits &&
               ((p->flags & RT_FLAG_REGROUP) || (regroup_wanted && h->accel_stats.nodes >= POOL_MIN_NODES))) {
        // behind a tree the pool of paths per wave pays where traversals are long (measured: +14 % at 466 k
        // nodes, -2 % at 232); RT_FLAG_REGROUP asks for it on any tree
#define RT_POOL(MODE)                                                                                                        \
    (h->accel == ACCEL_BVH4Q ? (extras  ? launch_paths_pool<MODE, true, true>(h, p, film, total)                               \
                                : plain ? launch_paths_pool<MODE, true, false, true>(h, p, film, total)                        \
                                        : launch_paths_pool<MODE, true, false>(h, p, film, total))                             \
                             : (extras  ? launch_paths_pool<MODE, false, true>(h, p, film, total)                              \
                                : plain ? launch_paths_pool<MODE, false, false, true>(h, p, film, total)                       \
                                        : launch_paths_pool<MODE, false, false>(h, p, film, total)))
        rc = p->mode == RT_MODE_REF_EXACT ? RT_POOL(RT_MODE_REF_EXACT) : RT_POOL(RT_MODE_BOOK3);
#undef RT_POOL
    } else {
#define RT_PATHS(MODE)                                                                                    \
    (h->accel == ACCEL_FLAT   ? launch_paths<MODE, ACCEL_FLAT>(h, dc, p, film, total)                        \
     : h->accel == ACCEL_BVH4 ? (extras  ? (simple ? launch_paths<MODE, ACCEL_BVH4, true, false, true>(h, dc, p, film, total) : launch_paths<MODE, ACCEL_BVH4, true>(h, dc, p, film, total)) \
                                 : plain ? launch_paths<MODE, ACCEL_BVH4, false, true>(h, dc, p, film, total)         \
                                         : launch_paths<MODE, ACCEL_BVH4, false>(h, dc, p, film, total))              \
     : h->accel == ACCEL_BVH4Q ? (extras  ? (simple ? launch_paths<MODE, ACCEL_BVH4Q, true, false, true>(h, dc, p, film, total) : launch_paths<MODE, ACCEL_BVH4Q, true>(h, dc, p, film, total)) \
                                  : plain ? launch_paths<MODE, ACCEL_BVH4Q, false, true>(h, dc, p, film, total)       \
                                          : launch_paths<MODE, ACCEL_BVH4Q, false>(h, dc, p, film, total))            \
                              : launch_paths<MODE, ACCEL_BVH2>(h, dc, p, film, total))
    rc = p->mode == RT_MODE_REF_EXACT ? RT_PATHS(RT_MODE_REF_EXACT) : RT_PATHS(RT_MODE_BOOK3);
#undef RT_PATHS
    }
    if (rc) return rc;
    h->last_variant = (h->stats.regrouped ? RT_VARIANT_REGROUPED : 0u) | (h->accel != ACCEL_FLAT && extras ? RT_VARIANT_EXTRAS : 0u) |
                      (h->accel != ACCEL_FLAT && plain ? RT_VARIANT_PLAIN : 0u) | (accel_is_wide(h->accel) && !h->stats.regrouped && simple ? RT_VARIANT_SIMPLE : 0u) | variant_flat;
    RT_HIP(hipEventRecord(h->ev_end, h->stream));
    RT_HIP(hipMemcpyAsync(&h->h_ctrl[0], h->d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
    RT_HIP(hipStreamSynchronize(h->stream));
    const Ctrl &fin = h->h_ctrl[0];
    h->stats.segments = fin.segments;
    h->stats.camera_rays = fin.camera_rays;
    h->stats.dropped = fin.dropped;
    float ms = 0.f;
    RT_HIP(hipEventElapsedTime(&ms, h->ev_begin, h->ev_end));
    h->stats.render_ms = ms;
    drain_timers(h);
    return 0;
}

// ------------------------------------------------------------------ RCCL, loaded on first use
// The film reduce is the only collective of the path.  librccl.so.1 (570 MB) is opened when a group
// is formed, so a single-GPU user never loads it; in a process that already holds it (PyTorch) the
// same soname resolves to that instance.
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
std::mutex g_rccl_mutex;
Rccl g_rccl;
const Rccl *rccl(std::string &err) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return &g_rccl;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        const char *why = dlerror();
        err = std::string("cannot load librccl.so.1: ") + (why ? why : "unknown reason");
        return nullptr;
    }
    Rccl r;
    bool ok = true;
    auto sym = [&](const char *name, auto &fn) {
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(lib, name));
        if (!fn) ok = false, err = std::string("librccl.so.1 has no ") + name;
    };
    sym("ncclGetUniqueId", r.GetUniqueId);
    sym("ncclCommInitRank", r.CommInitRank);
    sym("ncclCommInitAll", r.CommInitAll);
    sym("ncclCommDestroy", r.CommDestroy);
    sym("ncclAllReduce", r.AllReduce);
    sym("ncclGroupStart", r.GroupStart);
    sym("ncclGroupEnd", r.GroupEnd);
    sym("ncclGetErrorString", r.GetErrorString);
    if (!ok) {
        dlclose(lib);
        return nullptr;
    }
    r.lib = lib;
    g_rccl = r;
    return &g_rccl;
}
std::string rccl_what(const Rccl *r, const char *call, ncclResult_t rc) {
    return std::string(call) + ": " + r->GetErrorString(rc);
}
// groups formed by rt_comm_init_all get a number, so that rt_film_allreduce can tell that the
// handles it is given are the members of ONE group; 0 = a group of one handle per process
unsigned long long g_next_group = 1;

} // namespace

extern "C" {

#ifdef RT_REGROUP_STATS
// instrumented builds only: turns and lanes per step of k_paths_regroup since the last call
int rt_debug_regroup_stats(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rg_stats), sizeof(unsigned long long) * 3 * RG_STATS) != hipSuccess) return -1;
    unsigned long long zero[3 * RG_STATS] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rg_stats), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }
uint32_t rt_last_kernel_variant(const rt_handle *h) { return h ? h->last_variant : 0u; }

int rt_device_count(int *n) {
    rt_handle *h = nullptr;
    if (!n) return fail(h, "rt_device_count: null argument");
    RT_HIP(hipGetDeviceCount(n));
    return 0;
}

int rt_create(int device_id, rt_handle **out) {
    rt_handle *h = nullptr;
    if (!out) return fail(h, "rt_create: null argument");
    *out = nullptr;
    int n = 0;
    RT_HIP(hipGetDeviceCount(&n));
    if (n <= 0) return fail(h, "rt_create: no HIP device visible; this backend has no CPU path");
    if (device_id < 0 || device_id >= n) return fail(h, "rt_create: device id out of range");
    RT_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(h, std::string("rt_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    rt_handle *nh = new rt_handle();
    nh->device = device_id;
    h = nh;
    hipError_t e = hipStreamCreateWithFlags(&nh->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&nh->d_ctrl, sizeof(Ctrl));
    if (e == hipSuccess) e = hipMalloc((void **)&nh->d_sc, sizeof(DScene));
    if (e == hipSuccess) e = hipMalloc((void **)&nh->d_cam, sizeof(DCamera));
    if (e == hipSuccess) e = hipHostMalloc((void **)&nh->h_ctrl, 2 * sizeof(Ctrl), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreate(&nh->ev_begin);
    if (e == hipSuccess) e = hipEventCreate(&nh->ev_end);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&nh->ev_poll[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&nh->ev_poll[1], hipEventDisableTiming);
    if (e != hipSuccess) {
        g_create_error = std::string("rt_create: ") + hipGetErrorString(e);
        rt_destroy(nh);
        return -1;
    }
    nh->own_stream = true;
    if (scratch_for_device(nh, device_id, &nh->noalloc_scratch)) {
        g_create_error = "rt_create: " + nh->err;
        rt_destroy(nh);
        return -1;
    }
    *out = nh;
    return 0;
}

void rt_destroy(rt_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm) (void)rt_comm_destroy(h);
    free_scene(h);
    free_queue(h);
    if (h->film && h->film_owned) (void)hipFree(h->film);
    if (h->stage) (void)hipFree(h->stage);
    if (h->spill) (void)hipFree(h->spill);
    if (h->d_ctrl) (void)hipFree(h->d_ctrl);
    if (h->d_sc) (void)hipFree(h->d_sc);
    if (h->d_cam) (void)hipFree(h->d_cam);
    if (h->h_ctrl) (void)hipHostFree(h->h_ctrl);
    for (auto &p : h->ev_pool) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (hipEvent_t e : {h->ev_begin, h->ev_end, h->ev_poll[0], h->ev_poll[1]})
        if (e) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *rt_last_error(rt_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int rt_set_stream(rt_handle *h, void *hip_stream) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    RT_HIP(hipStreamSynchronize(h->stream));
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    h->own_stream = false;
    if (hip_stream) {
        h->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        RT_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return 0;
}

int rt_synchronize(rt_handle *h) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    RT_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int rt_upload_scene(rt_handle *h, const rt_scene_desc *s) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (validate_scene(h, s)) return -1;
    RT_HIP(hipStreamSynchronize(h->stream));
    // The scene in place stays until the new one is complete: a rejected upload leaves the handle
    // as it was.
    std::vector<void *> old_allocs;
    old_allocs.swap(h->scene_allocs);
    struct {
        bool flat;
        int accel;
        uint32_t stack4, lds_nodes4, n_box_leaves;
        accel::Stats stats;
    } const keep{h->flat, h->accel, h->stack4, h->lds_nodes4, h->n_box_leaves, h->accel_stats};
    auto rollback = [&]() {
        for (void *p : h->scene_allocs) (void)hipFree(p);
        h->scene_allocs.swap(old_allocs);
        h->flat = keep.flat, h->accel = keep.accel, h->stack4 = keep.stack4, h->lds_nodes4 = keep.lds_nodes4, h->accel_stats = keep.stats;
        h->n_box_leaves = keep.n_box_leaves;
        return -1;
    };
    DScene d{};
    static_assert(sizeof(rt_quad) == 96 && sizeof(rt_sphere) == 48 && sizeof(rt_node) == 64, "record sizes");
    static_assert(sizeof(rt_material) == 32 && sizeof(rt_texture) == 32 && sizeof(rt_medium) == 32, "record sizes");
    const rt_quad *dq;
    const rt_sphere *ds;
    const float *dnv;
    const Node4 *dn4 = nullptr;
    const Node4Q *dn4q = nullptr;
    const WideBox *dwb = nullptr;
    // device copy of the materials: a SolidColor texture is folded into the material record
    // (albedo + pad = 1) so that shading such a surface does not chase the texture record
    std::vector<rt_material> mats(s->materials, s->materials + s->n_materials);
    bool plain_materials = true;
    for (rt_material &m : mats) {
        m.pad = 0;
        const bool textured = m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_LIGHT || m.type == RT_MAT_ISOTROPIC;
        if (textured && s->textures[m.tex].type == RT_TEX_SOLID) {
            std::memcpy(m.albedo, s->textures[m.tex].color, sizeof(m.albedo));
            m.pad = 1;
        }
        if ((textured && !m.pad) || m.type == RT_MAT_ISOTROPIC) plain_materials = false;
    }
    // A world that is one list of primitives takes the stack-free scan (rt_flat.h): its quads packed
    // in list order for scalar loads, boxes and a room as slab tests, everything else by reference.
    flat::Plan fp;
    {
        const std::string err = flat::plan(s, fp);
        if (!err.empty()) {
            fail(h, "rt_upload_scene: " + err);
            return rollback();
        }
    }
    std::vector<float> &flat_quads = fp.quads, &flat_pairs = fp.pairs;
    std::vector<uint32_t> &flat_other = fp.other;
    // one array for the scan's second half: the rectangles' records, padded to whole box records, then the boxes'
    std::vector<float> flat_boxes(fp.rects);
    flat_boxes.resize((flat_boxes.size() + 31) / 32 * 32, 0.0f);
    flat_boxes.insert(flat_boxes.end(), fp.boxes.begin(), fp.boxes.end());
    // the light list's primitives, one record per light in list order (light_random in rt_kernels.h)
    std::vector<float> light_recs(size_t(s->n_lights) * 32, 0.0f);
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        float *L = &light_recs[size_t(i) * 32];
        const uint32_t lr = s->light_refs[i], idx = lr & RT_REF_INDEX_MASK;
        if ((lr >> RT_REF_KIND_SHIFT) == RT_KIND_QUAD) {
            const rt_quad &q = s->quads[idx];
            for (int a = 0; a < 3; ++a) L[a] = q.q[a], L[4 + a] = q.n[a], L[8 + a] = q.avec[a], L[12 + a] = q.bvec[a], L[16 + a] = q.u[a], L[20 + a] = q.v[a];
            L[7] = q.area;
        } else {
            const rt_sphere &sp = s->spheres[idx];
            for (int a = 0; a < 3; ++a) L[a] = sp.c0[a];
            L[3] = sp.radius;
        }
    }
    const float *d_light_recs = nullptr;
    // device copy of the spheres: pad = 1 marks a sphere of a flat world that nothing reaches into
    std::vector<rt_sphere> spheres(s->spheres, s->spheres + s->n_spheres);
    for (rt_sphere &sp : spheres) sp.pad = 0;
    for (uint32_t r : fp.alone_spheres) spheres[r & RT_REF_INDEX_MASK].pad = 1;
    h->flat = fp.flat;
    // structure the kernels will search
    std::vector<Node4> nodes4;
    std::vector<Node4Q> nodes4q;
    std::vector<WideBox> wide_boxes; // the tree's leaves of kind RT_KIND_BOX
    std::vector<uint32_t> post;      // primitives that enclose the scene: tested after the walk (test_post)
    h->stack4 = h->lds_nodes4 = 0;
    h->accel_stats = accel::Stats();
    if (h->flat) {
        h->accel = ACCEL_FLAT;
    } else if (h->width == 4) {
        uint32_t root4 = 0;
        std::string err = accel::build_wide_bvh(s, h->build, nodes4, root4, h->accel_stats, h->box_leaves ? &wide_boxes : nullptr, nullptr,
                                                h->post_prims ? &post : nullptr);
        if (!err.empty()) {
            fail(h, "rt_upload_scene: " + err);
            return rollback();
        }
        // one LDS column per lane, 1 KB per entry per block: up to 40 entries leave 4 blocks per CU
        // (160 KB of LDS), deeper trees run at lower occupancy, 64 entries is the 64 KB block limit
        if (h->accel_stats.stack_need <= 64 && nodes4.size() < (1u << 25)) { // byte offsets of nodes stay 32-bit
            // byte boxes pay once the tree no longer lives in LDS and L2 (measured: +5 % from 16 k nodes, +26 %
            // at 466 k; the decode costs 3-7 % on trees of hundreds of nodes, which are issue-bound)
            const bool byte_boxes = h->boxes == rt_handle::BOXES_BYTE || (h->boxes == rt_handle::BOXES_AUTO && nodes4.size() >= 8192);
            h->accel = byte_boxes ? ACCEL_BVH4Q : ACCEL_BVH4;
            if (h->accel == ACCEL_BVH4Q) {
                accel::quantise_nodes(nodes4, nodes4q);
                const std::string bad = accel::verify_quantised(nodes4, nodes4q);
                if (!bad.empty()) {
                    fail(h, "rt_upload_scene: " + bad);
                    return rollback();
                }
            }
            h->stack4 = std::max(8u, (h->accel_stats.stack_need + 3u) & ~3u);
            // what is left of a quarter of the CU's 160 KB (four resident blocks) holds the top of the tree
            // (measured: 52 or 64 KB per block stage more nodes but lose a resident block: slower)
            const size_t budget = 40 * 1024, stack_bytes = size_t(h->stack4) * BLOCK * sizeof(uint32_t);
            h->lds_nodes4 = stack_bytes < budget ? uint32_t(std::min<size_t>(nodes4.size(), (budget - stack_bytes) / h->node4_bytes())) : 0u;
        } else {
            nodes4.clear(); // too deep for the LDS stack of the wide traversal: binary nodes as given
            wide_boxes.clear();
            post.clear();
            h->accel = ACCEL_BVH2;
        }
    } else {
        h->accel = ACCEL_BVH2;
    }
    if (h->accel == ACCEL_BVH2) {
        // the binary traversal drops pushes beyond its fixed LDS stack: refuse instead of missing hits
        // (recomputed here; the descriptor's max_stack is the caller's word)
        const uint32_t need = std::max(binary_stack_need(s), s->max_stack);
        if (need > uint32_t(STACK_DEPTH)) {
            fail(h, "rt_upload_scene: scene needs a deeper traversal stack than the kernels provide (" + std::to_string(need) + " > " +
                        std::to_string(STACK_DEPTH) + ")");
            return rollback();
        }
    }
    d.n_post = uint32_t(post.size());
    for (size_t i = 0; i < post.size() && i < 2; ++i) d.post_ref[i] = post[i];
    d.n_nodes4 = uint32_t(nodes4.size());
    d.stack4 = h->stack4;
    d.lds_nodes4 = h->lds_nodes4;
    d.n_flat_pairs = uint32_t(flat_pairs.size() / 32);
    d.n_flat_quads = fp.n_quads; // (the array is padded with records that cannot hit)
    d.n_flat_other = uint32_t(flat_other.size());
    d.n_flat_spheres = uint32_t(fp.spheres.size() / 16);
    d.n_flat_boxes = uint32_t(fp.boxes.size() / 32);
    d.n_flat_rects = uint32_t(fp.rects.size() / 8), d.n_flat_rects_x = fp.n_rects_x, d.n_flat_rects_y = fp.n_rects_y;
    d.n_flat_boxes_aligned = fp.n_boxes_aligned, d.n_flat_boxes_upright = fp.n_boxes_upright;
    if (upload(h, flat_boxes.data(), flat_boxes.size(), &d.flat_boxes) || upload(h, s->nodes, s->n_nodes, &d.nodes) || upload(h, s->refs, s->n_refs, &d.refs) ||
        upload(h, s->quads, s->n_quads, &dq) || upload(h, spheres.data(), spheres.size(), &ds) ||
        upload(h, flat_quads.data(), flat_quads.size(), &d.flat_quads) || upload(h, flat_pairs.data(), flat_pairs.size(), &d.flat_pairs) || (h->accel == ACCEL_BVH4Q ? upload(h, nodes4q.data(), nodes4q.size(), &dn4q) : upload(h, nodes4.data(), nodes4.size(), &dn4)) ||
        upload(h, wide_boxes.data(), wide_boxes.size(), &dwb) ||
        upload(h, flat_other.data(), flat_other.size(), &d.flat_other) || upload(h, fp.spheres.data(), fp.spheres.size(), &d.flat_spheres) ||
        upload(h, s->media, s->n_media, &d.media) || upload(h, mats.data(), mats.size(), &d.mats) ||
        upload(h, s->textures, s->n_textures, &d.texs) || upload(h, s->light_refs, s->n_lights, &d.light_refs) || upload(h, light_recs.data(), light_recs.size(), &d_light_recs) ||
        upload(h, s->texels, s->n_texel_bytes, &d.texels) || upload(h, s->noise_vec, size_t(s->n_noise) * 1024, &dnv) ||
        upload(h, s->noise_perm, size_t(s->n_noise) * 768, &d.noise_perm))
        return rollback();
    d.quads = reinterpret_cast<const float4 *>(dq);
    d.light_recs = reinterpret_cast<const float4 *>(d_light_recs);
    d.spheres = reinterpret_cast<const float4 *>(ds);
    d.noise_vec = reinterpret_cast<const float4 *>(dnv);
    d.nodes4 = h->accel == ACCEL_BVH4Q ? reinterpret_cast<const float4 *>(dn4q) : reinterpret_cast<const float4 *>(dn4);
    d.wide_boxes = reinterpret_cast<const float4 *>(dwb);
    d.root = s->root;
    d.n_lights = s->n_lights;
    d.lights_kind = s->lights_kind;
    d.n_nodes = s->n_nodes;
    {
        hipError_t e = hipMemcpyAsync(h->d_sc, &d, sizeof(DScene), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { // d_sc may hold the new descriptor already: no scene rather than a mixed one
            fail(h, std::string("rt_upload_scene: ") + hipGetErrorString(e));
            rollback();
            free_scene(h);
            return -1;
        }
    }
    for (void *p : old_allocs) (void)hipFree(p);
    h->sc = d;
    h->have_scene = true;
    h->has_media = s->n_media ? 1u : 0u;
    h->plain_materials = plain_materials;
    h->n_box_leaves = uint32_t(wide_boxes.size());
    h->n_quads = s->n_quads, h->n_spheres = s->n_spheres, h->n_materials = s->n_materials;
    h->alone_spheres = 0;
    for (const rt_sphere &sp : spheres) h->alone_spheres += sp.pad;
    h->n_prims = s->n_prims;
    h->n_textures = s->n_textures;
    h->has_specular = false;
    h->has_motion = false;
    for (uint32_t i = 0; i < s->n_spheres; ++i)
        if (s->spheres[i].moving || s->spheres[i].cv[0] != 0.f || s->spheres[i].cv[1] != 0.f || s->spheres[i].cv[2] != 0.f) h->has_motion = true;
    for (uint32_t i = 0; i < s->n_materials; ++i)
        if (s->materials[i].type == RT_MAT_METAL || s->materials[i].type == RT_MAT_DIELECTRIC) h->has_specular = true;
    h->prim_to_ref.assign(s->n_prims, RT_REF_NONE);
    auto note = [&](uint32_t id, uint32_t ref) {
        if (id < s->n_prims) h->prim_to_ref[id] = ref;
    };
    for (uint32_t i = 0; i < s->n_quads; ++i) note(s->quads[i].prim_id, (RT_KIND_QUAD << RT_REF_KIND_SHIFT) | i);
    for (uint32_t i = 0; i < s->n_spheres; ++i) note(s->spheres[i].prim_id, (RT_KIND_SPHERE << RT_REF_KIND_SHIFT) | i);
    for (uint32_t i = 0; i < s->n_media; ++i) note(s->media[i].prim_id, (RT_KIND_MEDIUM << RT_REF_KIND_SHIFT) | i);
    return 0;
}

int rt_set_accel(rt_handle *h, uint32_t builder, uint32_t width) {
    if (!h) return -1;
    if ((builder & RT_BUILD_FP32_BOXES) && (builder & RT_BUILD_BYTE_BOXES)) return fail(h, "rt_set_accel: fp32 boxes or byte boxes, not both");
    const int boxes = (builder & RT_BUILD_FP32_BOXES) ? rt_handle::BOXES_FP32 : (builder & RT_BUILD_BYTE_BOXES) ? rt_handle::BOXES_BYTE : rt_handle::BOXES_AUTO;
    const bool box_leaves = !(builder & RT_BUILD_NO_BOX_LEAVES), post_prims = !(builder & RT_BUILD_NO_POST_TESTS);
    builder &= ~(RT_BUILD_FP32_BOXES | RT_BUILD_BYTE_BOXES | RT_BUILD_NO_BOX_LEAVES | RT_BUILD_NO_POST_TESTS);
    if (builder > RT_BUILD_AUTO) return fail(h, "rt_set_accel: unknown builder");
    if (width != 2 && width != 4) return fail(h, "rt_set_accel: width must be 2 or 4");
    if (width == 2 && builder != RT_BUILD_REFERENCE)
        return fail(h, "rt_set_accel: the binary traversal walks the descriptor's own nodes; regrouping needs width 4");
    h->build = builder;
    h->width = width;
    h->boxes = boxes;
    h->box_leaves = box_leaves;
    h->post_prims = post_prims;
    return 0;
}

int rt_accel_selftest(const rt_scene_desc *scene, uint32_t builder, rt_accel_info *out, char *err, uint32_t err_len) {
    auto say = [&](const std::string &m) {
        if (err && err_len) std::snprintf(err, err_len, "%s", m.c_str());
        return -1;
    };
    if (!scene || builder > RT_BUILD_AUTO) return say("rt_accel_selftest: bad argument");
    rt_handle tmp; // validate_scene only records its message in the handle
    if (validate_scene(&tmp, scene)) return say(tmp.err);
    std::vector<Node4> nodes;
    std::vector<WideBox> boxes;
    std::vector<uint32_t> box_faces;
    uint32_t root = 0;
    accel::Stats st;
    std::vector<uint32_t> post;
    std::string e = accel::build_wide_bvh(scene, builder, nodes, root, st, &boxes, &box_faces, &post);
    if (e.empty()) e = accel::verify_wide_bvh(scene, nodes, boxes, box_faces, post);
    if (!e.empty()) return say("rt_accel_selftest: " + e);
    if (out) {
        rt_accel_info a{};
        a.kind = uint32_t(ACCEL_BVH4);
        a.nodes = st.nodes, a.depth = st.depth, a.stack_entries = st.stack_need, a.cost = st.cost;
        // what the upload would make of the scene as a flat world (0s unless it is one list of primitives)
        flat::Plan fp;
        e = flat::plan(scene, fp);
        if (!e.empty()) return say("rt_accel_selftest: " + e);
        a.flat_quads = fp.n_quads, a.flat_boxes = uint32_t(fp.boxes.size() / 32), a.flat_alone_spheres = uint32_t(fp.alone_spheres.size());
        a.flat_boxes_aligned = fp.n_boxes_aligned, a.flat_boxes_upright = fp.n_boxes_upright;
        a.flat_rects = uint32_t(fp.rects.size() / 8);
        a.box_leaves = uint32_t(boxes.size()), a.leaves = st.leaves, a.post_tests = uint32_t(post.size()), a.walk_cost = st.walk;
        a.node_bytes = uint32_t(sizeof(Node4Q));
        std::vector<Node4Q> q;
        accel::quantise_nodes(nodes, q);
        e = accel::verify_quantised(nodes, q);
        if (!e.empty()) return say("rt_accel_selftest: " + e);
        e = write_sized(a, out, "rt_accel_info");
        if (!e.empty()) return say("rt_accel_selftest: " + e);
    }
    return 0;
}

int rt_get_accel_info(rt_handle *h, rt_accel_info *out) {
    if (!h || !out) return -1;
    if (!h->have_scene) return fail(h, "rt_get_accel_info: no scene uploaded");
    const bool wide = accel_is_wide(h->accel);
    rt_accel_info a{};
    a.kind = wide ? uint32_t(ACCEL_BVH4) : uint32_t(h->accel);
    a.nodes = wide ? h->accel_stats.nodes : (h->accel == ACCEL_BVH2 ? h->sc.n_nodes : 0u);
    a.depth = wide ? h->accel_stats.depth : 0u;
    a.stack_entries = wide ? h->stack4 : (h->accel == ACCEL_BVH2 ? uint32_t(STACK_DEPTH) : 0u);
    a.cost = wide ? h->accel_stats.cost : 0.0;
    a.node_bytes = wide ? uint32_t(h->node4_bytes()) : (h->accel == ACCEL_BVH2 ? uint32_t(sizeof(rt_node)) : 0u);
    a.flat_quads = h->accel == ACCEL_FLAT ? h->sc.n_flat_quads : 0u;
    a.flat_boxes = h->accel == ACCEL_FLAT ? h->sc.n_flat_boxes : 0u;
    a.flat_alone_spheres = h->alone_spheres;
    a.flat_boxes_aligned = h->accel == ACCEL_FLAT ? h->sc.n_flat_boxes_aligned : 0u;
    a.flat_boxes_upright = h->accel == ACCEL_FLAT ? h->sc.n_flat_boxes_upright : 0u;
    a.flat_rects = h->accel == ACCEL_FLAT ? h->sc.n_flat_rects : 0u;
    a.box_leaves = wide ? h->n_box_leaves : 0u, a.leaves = wide ? h->accel_stats.leaves : 0u;
    a.post_tests = wide ? h->sc.n_post : 0u, a.walk_cost = wide ? h->accel_stats.walk : 0.0;
    const std::string bad = write_sized(a, out, "rt_accel_info");
    return bad.empty() ? 0 : fail(h, "rt_get_accel_info: " + bad);
}

int rt_film_clear(rt_handle *h) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (!h->film) return 0;
    RT_HIP(hipMemsetAsync(h->film, 0, size_t(h->film_w) * h->film_h * 3 * sizeof(float), h->stream));
    return 0;
}

int rt_film_resize(rt_handle *h, uint32_t w, uint32_t hgt) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (w == 0 || hgt == 0) return fail(h, "rt_film_resize: empty film");
    if (h->film && !h->film_owned) {
        h->film = nullptr;
        h->film_w = h->film_h = 0;
    }
    if (ensure_film(h, w, hgt)) return -1;
    return rt_film_clear(h);
}

int rt_film_bind(rt_handle *h, void *device_ptr, uint32_t w, uint32_t hgt) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (!device_ptr || w == 0 || hgt == 0) return fail(h, "rt_film_bind: null pointer or empty film");
    if ((reinterpret_cast<uintptr_t>(device_ptr) & 3u) != 0) return fail(h, "rt_film_bind: pointer not 4-byte aligned");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, device_ptr) != hipSuccess || attr.type != hipMemoryTypeDevice)
        return fail(h, "rt_film_bind: not a device pointer");
    if (attr.device != h->device) return fail(h, "rt_film_bind: pointer lives on another device");
    RT_HIP(hipStreamSynchronize(h->stream));
    if (h->film && h->film_owned) (void)hipFree(h->film);
    h->film = static_cast<float *>(device_ptr);
    h->film_owned = false;
    h->film_w = w;
    h->film_h = hgt;
    return 0;
}

int rt_film_device_ptr(rt_handle *h, void **device_ptr, uint64_t *n_floats) {
    if (!h || !device_ptr || !n_floats) return -1;
    *device_ptr = h->film;
    *n_floats = uint64_t(h->film_w) * h->film_h * 3;
    return 0;
}

int rt_film_download(rt_handle *h, float *rgb, float scale) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (!h->film) return fail(h, "rt_film_download: no film");
    if (!rgb) return fail(h, "rt_film_download: null destination");
    uint32_t npix = h->film_w * h->film_h;
    void *stage = nullptr;
    if (ensure_stage(h, size_t(npix) * 3 * sizeof(float), &stage)) return -1;
    float *tmp = static_cast<float *>(stage);
    k_film_pack<<<(npix + 255) / 256, 256, 0, h->stream>>>(h->film, h->film + npix, h->film + 2 * size_t(npix), npix, scale, tmp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(rgb, tmp, size_t(npix) * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, std::string("rt_film_download: ") + hipGetErrorString(e));
    return 0;
}

int rt_film_upload(rt_handle *h, const float *rgb, uint32_t w, uint32_t hgt) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (!rgb || w == 0 || hgt == 0) return fail(h, "rt_film_upload: null source or empty film");
    if (h->film && !h->film_owned && (w != h->film_w || hgt != h->film_h))
        return fail(h, "rt_film_upload: size differs from the bound film");
    if (ensure_film(h, w, hgt)) return -1;
    const uint32_t npix = w * hgt;
    void *stage = nullptr;
    if (ensure_stage(h, size_t(npix) * 3 * sizeof(float), &stage)) return -1;
    float *tmp = static_cast<float *>(stage);
    hipError_t e = hipMemcpyAsync(tmp, rgb, size_t(npix) * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        k_film_unpack<<<(npix + 255) / 256, 256, 0, h->stream>>>(tmp, npix, h->film, h->film + npix, h->film + 2 * size_t(npix));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, std::string("rt_film_upload: ") + hipGetErrorString(e));
    return 0;
}

namespace {
// film sums -> packed bytes in `packed` (device) -> rgb8 (host)
int quantize_into(rt_handle *h, const char *who, uint32_t *packed, uint8_t *rgb8, double scale, uint32_t gamma) {
    const uint32_t npix = h->film_w * h->film_h;
    const uint32_t groups = (npix + 3u) / 4u;
    k_film_quantize<<<(groups + 255) / 256, 256, 0, h->stream>>>(h->film, h->film + npix, h->film + 2 * size_t(npix), npix, scale,
                                                                gamma, packed);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(rgb8, packed, size_t(npix) * 3, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, std::string(who) + ": " + hipGetErrorString(e));
    return 0;
}
} // namespace

int rt_film_quantize(rt_handle *h, uint8_t *rgb8, double scale, uint32_t gamma) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (!h->film) return fail(h, "rt_film_quantize: no film");
    if (!rgb8) return fail(h, "rt_film_quantize: null destination");
    const uint32_t groups = (h->film_w * h->film_h + 3u) / 4u;
    void *stage = nullptr;
    if (ensure_stage(h, size_t(groups) * 12, &stage)) return -1;
    return quantize_into(h, "rt_film_quantize", static_cast<uint32_t *>(stage), rgb8, scale, gamma);
}

int rt_film_quantize_noalloc(rt_handle *h, uint8_t *rgb8, double scale, uint32_t gamma) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (!h->film) return fail(h, "rt_film_quantize_noalloc: no film");
    if (!rgb8) return fail(h, "rt_film_quantize_noalloc: null destination");
    if (uint64_t(h->film_w) * h->film_h > NOALLOC_MAX_PIXELS)
        return fail(h, "rt_film_quantize_noalloc: film of " + std::to_string(uint64_t(h->film_w) * h->film_h) +
                           " pixels exceeds the fixed scratch of " + std::to_string(NOALLOC_MAX_PIXELS) +
                           " pixels (3840 x 2160); use rt_film_quantize");
    // the kernel overwrites every word of the scratch that the copy reads: nothing to clear.  The lock
    // covers the launch, the copy and the stream synchronise (quantize_into returns when the bytes are home).
    std::lock_guard<std::mutex> lock(g_scratch_in_use[h->device]);
    return quantize_into(h, "rt_film_quantize_noalloc", h->noalloc_scratch, rgb8, scale, gamma);
}

int rt_comm_init_all(rt_handle *const *handles, int n) {
    if (!handles || n < 1 || !handles[0]) return -1;
    rt_handle *h = handles[0];
    std::vector<int> devs;
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return fail(h, "rt_comm_init_all: null handle in the list");
        if (handles[i]->comm) return fail(h, "rt_comm_init_all: handle " + std::to_string(i) + " already belongs to a group (rt_comm_destroy first)");
        for (int d : devs)
            if (d == handles[i]->device) return fail(h, "rt_comm_init_all: two handles on GPU " + std::to_string(d) + " (RCCL takes one rank per GPU)");
        devs.push_back(handles[i]->device);
    }
    std::string err;
    const Rccl *r = rccl(err);
    if (!r) return fail(h, "rt_comm_init_all: " + err);
    std::vector<ncclComm_t> comms(n, nullptr);
    const ncclResult_t rc = r->CommInitAll(comms.data(), n, devs.data());
    if (rc != ncclSuccess) return fail(h, "rt_comm_init_all: " + rccl_what(r, "ncclCommInitAll", rc));
    unsigned long long group;
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        group = g_next_group++;
    }
    for (int i = 0; i < n; ++i) handles[i]->comm = comms[i], handles[i]->comm_rank = i, handles[i]->comm_size = n, handles[i]->comm_group = group;
    return 0;
}

int rt_comm_unique_id(void *id, char *err, uint32_t err_len) {
    auto say = [&](const std::string &m) {
        if (err && err_len) std::snprintf(err, err_len, "%s", m.c_str());
        return -1;
    };
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "the id is handed over as RT_COMM_ID_BYTES opaque bytes");
    if (!id) return say("rt_comm_unique_id: null argument");
    std::string e;
    const Rccl *r = rccl(e);
    if (!r) return say("rt_comm_unique_id: " + e);
    ncclUniqueId uid;
    const ncclResult_t rc = r->GetUniqueId(&uid);
    if (rc != ncclSuccess) return say("rt_comm_unique_id: " + rccl_what(r, "ncclGetUniqueId", rc));
    std::memcpy(id, &uid, sizeof(uid));
    return 0;
}

int rt_comm_init_rank(rt_handle *h, const void *id, int rank, int n_ranks) {
    if (!h) return -1;
    if (!id) return fail(h, "rt_comm_init_rank: null id");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(h, "rt_comm_init_rank: rank outside 0..n_ranks");
    if (h->comm) return fail(h, "rt_comm_init_rank: the handle already belongs to a group (rt_comm_destroy first)");
    std::string err;
    const Rccl *r = rccl(err);
    if (!r) return fail(h, "rt_comm_init_rank: " + err);
    RT_HIP(hipSetDevice(h->device)); // the communicator binds to the current device
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r->CommInitRank(&comm, n_ranks, uid, rank);
    if (rc != ncclSuccess) return fail(h, "rt_comm_init_rank: " + rccl_what(r, "ncclCommInitRank", rc));
    h->comm = comm, h->comm_rank = rank, h->comm_size = n_ranks, h->comm_group = 0;
    return 0;
}

int rt_comm_destroy(rt_handle *h) {
    if (!h) return -1;
    if (!h->comm) return 0;
    std::string err;
    const Rccl *r = rccl(err);
    if (!r) return fail(h, "rt_comm_destroy: " + err);
    RT_HIP(hipSetDevice(h->device));
    RT_HIP(hipStreamSynchronize(h->stream)); // a reduce in flight still uses the communicator
    const ncclResult_t rc = r->CommDestroy(h->comm);
    h->comm = nullptr, h->comm_rank = h->comm_size = 0, h->comm_group = 0;
    if (rc != ncclSuccess) return fail(h, "rt_comm_destroy: " + rccl_what(r, "ncclCommDestroy", rc));
    return 0;
}

int rt_comm_info(rt_handle *h, int *rank, int *n_ranks) {
    if (!h || !rank || !n_ranks) return -1;
    *rank = h->comm ? h->comm_rank : 0;
    *n_ranks = h->comm ? h->comm_size : 0;
    return 0;
}

int rt_film_allreduce(rt_handle *const *handles, int n) {
    if (!handles || n < 1 || !handles[0]) return -1;
    rt_handle *h = handles[0];
    for (int i = 0; i < n; ++i) {
        rt_handle *m = handles[i];
        if (!m) return fail(h, "rt_film_allreduce: null handle in the list");
        if (!m->comm) return fail(h, "rt_film_allreduce: handle " + std::to_string(i) + " is in no group (rt_comm_init_all / rt_comm_init_rank)");
        if (!m->film) return fail(h, "rt_film_allreduce: handle " + std::to_string(i) + " has no film");
        if (m->film_w != h->film_w || m->film_h != h->film_h) return fail(h, "rt_film_allreduce: the films differ in size");
        if (m->comm_group != h->comm_group) return fail(h, "rt_film_allreduce: the handles belong to different groups");
        for (int k = 0; k < i; ++k)
            if (handles[k] == m) return fail(h, "rt_film_allreduce: a handle is listed twice");
    }
    // every member this process holds must take part in the one grouped call, or the others wait forever
    const int local_members = h->comm_group ? h->comm_size : 1;
    if (n != local_members)
        return fail(h, "rt_film_allreduce: " + std::to_string(n) + " handles given, this process holds " + std::to_string(local_members) + " member(s) of the group");
    std::string err;
    const Rccl *r = rccl(err);
    if (!r) return fail(h, "rt_film_allreduce: " + err);
    const size_t count = size_t(h->film_w) * h->film_h * 3;
    ncclResult_t rc = r->GroupStart();
    if (rc != ncclSuccess) return fail(h, "rt_film_allreduce: " + rccl_what(r, "ncclGroupStart", rc));
    ncclResult_t first_bad = ncclSuccess;
    for (int i = 0; i < n; ++i) {
        rt_handle *m = handles[i];
        if (hipSetDevice(m->device) != hipSuccess) {
            first_bad = ncclUnhandledCudaError;
            break;
        }
        // in place, on the handle's stream: ordered behind the render that filled the film
        rc = r->AllReduce(m->film, m->film, count, ncclFloat32, ncclSum, m->comm, m->stream);
        if (rc != ncclSuccess) {
            first_bad = rc;
            break;
        }
    }
    rc = r->GroupEnd(); // always closed, or the thread stays inside a group
    if (first_bad != ncclSuccess) return fail(h, "rt_film_allreduce: " + rccl_what(r, "ncclAllReduce", first_bad));
    if (rc != ncclSuccess) return fail(h, "rt_film_allreduce: " + rccl_what(r, "ncclGroupEnd", rc));
    return 0;
}

int rt_get_stats(rt_handle *h, rt_stats *out) {
    if (!h || !out) return -1;
    const std::string bad = write_sized(h->stats, out, "rt_stats");
    return bad.empty() ? 0 : fail(h, "rt_get_stats: " + bad);
}

int rt_render(rt_handle *h, const rt_camera *cam, const rt_render_params *params) {
    if (!h) return -1;
    rt_render_params own;
    if (read_params(h, params, &own)) return -1;
    const rt_render_params *p = &own;
    if ((p->flags & RT_FLAG_REGROUP) && (p->flags & RT_FLAG_NO_REGROUP))
        return fail(h, "rt_render: RT_FLAG_REGROUP and RT_FLAG_NO_REGROUP are both set");
    RT_HIP(hipSetDevice(h->device));
    if (!h->have_scene) return fail(h, "rt_render: no scene uploaded");
    DCamera dc;
    uint32_t stratum_end;
    // Nothing moves: the time of a ray (camera.rs:429, the camera stream's last draw) decides nothing
    // and is not drawn -- behind a defocus disk that draw is a Philox block of its own.
    if (make_camera(h, cam, p, &dc, &stratum_end, !h->has_motion)) return -1;
    if (p->mode > RT_MODE_BOOK3) return fail(h, "rt_render: unknown estimator mode");
    if (p->mode == RT_MODE_REF_EXACT) {
        // Mirrors the reference's panics: Metal/Dielectric scatter is todo!() (material.rs:128,179)
        // and only Parallelogram implements pdf_value/random (hittable.rs:99-107, 377-395).
        if (h->has_specular)
            return fail(h, "rt_render: REF_EXACT cannot render Metal/Dielectric (the reference panics with todo!())");
        if (h->sc.lights_kind != RT_LIGHTS_SINGLE)
            return fail(h, "rt_render: REF_EXACT needs `lights` to be one Parallelogram (others are unimplemented!() in the reference)");
    }
    if (ensure_film(h, dc.W, dc.H)) return -1;

    const unsigned long long total = (unsigned long long)dc.npix * (stratum_end - dc.stratum_begin);
    std::memset(&h->stats, 0, sizeof(h->stats));
    if (total == 0) return 0;

    Ctrl init{};
    init.total_samples = total;
    h->h_ctrl[0] = init;
    RT_HIP(hipMemcpyAsync(h->d_ctrl, &h->h_ctrl[0], sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));
    RT_HIP(hipStreamSynchronize(h->stream)); // h_ctrl[0] is reused below

    const uint32_t k0 = uint32_t(p->seed), k1 = uint32_t(p->seed >> 32);
    const Film film{h->film, dc.npix};
    if (!(p->flags & RT_FLAG_WAVEFRONT)) return render_persistent(h, dc, p, film, total);
    h->last_variant = 0; // the ray-queue pipeline has one compilation

    // pool geometry: chunks of 512 rays, one wave each
    const uint32_t cap = 512;
    unsigned long long pool = p->pool_size ? p->pool_size : (1ull << 22);
    if (pool > (1ull << 26)) pool = 1ull << 26;
    if (pool > total) pool = total;
    uint32_t n_chunks = uint32_t((pool + cap - 1) / cap);
    n_chunks = (n_chunks + 3u) & ~3u;
    if (ensure_queue(h, n_chunks, cap)) return -1;

    RT_HIP(hipMemsetAsync(h->q.count, 0, n_chunks * sizeof(uint32_t), h->stream));
    const dim3 grid(n_chunks / (BLOCK / WAVE)), block(BLOCK);
    const bool timing = (p->flags & RT_FLAG_TIME_KERNELS) != 0;
    const uint32_t time_every = p->time_every ? p->time_every : 1u;
    Timer tm{h, false};
    const size_t dyn_lds = h->dyn_lds4(); // wide traversal: stack + staged nodes
    uint64_t iter_no = p->seed; // rotate which iterations are timed from render to render

    RT_HIP(hipEventRecord(h->ev_begin, h->stream));
    const int ITERS_PER_POLL = 8;
    int slot = 0;
    bool pending[2] = {false, false};
    size_t ev_mark[2] = {0, 0}; // event pairs in use when each poll event was recorded
    bool done = false;
    unsigned long long guard = 0;
    const unsigned long long max_iters = total / (size_t(n_chunks) * cap) * (dc.max_depth + 1ull) + 4ull * dc.max_depth + 64;
    while (!done) {
        for (int it = 0; it < ITERS_PER_POLL; ++it) {
            tm.on = timing && (iter_no++ % time_every) == 0;
            int e = tm.begin(3);
            k_scan<<<1, 1024, 0, h->stream>>>(h->q, h->d_ctrl);
            tm.end(e);
            e = tm.begin(0);
            k_generate<<<grid, block, 0, h->stream>>>(h->q, h->d_cam, k0, k1);
            tm.end(e);
            e = tm.begin(1);
            if (h->accel == ACCEL_FLAT)
                k_extend<ACCEL_FLAT><<<grid, block, 0, h->stream>>>(h->q, h->d_sc, k0, k1, h->has_media);
            else if (h->accel == ACCEL_BVH4)
                k_extend<ACCEL_BVH4><<<grid, block, dyn_lds, h->stream>>>(h->q, h->d_sc, k0, k1, h->has_media);
            else if (h->accel == ACCEL_BVH4Q)
                k_extend<ACCEL_BVH4Q><<<grid, block, dyn_lds, h->stream>>>(h->q, h->d_sc, k0, k1, h->has_media);
            else
                k_extend<ACCEL_BVH2><<<grid, block, 0, h->stream>>>(h->q, h->d_sc, k0, k1, h->has_media);
            tm.end(e);
            e = tm.begin(2);
            if (p->mode == RT_MODE_REF_EXACT)
                k_shade<RT_MODE_REF_EXACT><<<grid, block, 0, h->stream>>>(h->q, h->d_sc, h->d_cam, film, k0, k1, h->d_ctrl);
            else
                k_shade<RT_MODE_BOOK3><<<grid, block, 0, h->stream>>>(h->q, h->d_sc, h->d_cam, film, k0, k1, h->d_ctrl);
            tm.end(e);
            h->stats.launches[0]++, h->stats.launches[1]++, h->stats.launches[2]++, h->stats.launches[3]++;
        }
        RT_HIP(hipGetLastError());
        RT_HIP(hipMemcpyAsync(&h->h_ctrl[slot], h->d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
        RT_HIP(hipEventRecord(h->ev_poll[slot], h->stream));
        pending[slot] = true;
        ev_mark[slot] = h->ev_used;
        // look at the previous batch while this one runs
        int prev = slot ^ 1;
        if (pending[prev]) {
            RT_HIP(hipEventSynchronize(h->ev_poll[prev]));
            pending[prev] = false;
            // everything recorded before that poll event has finished: its timers go back to the pool
            const size_t done_pairs = ev_mark[prev];
            drain_timers(h, done_pairs);
            ev_mark[slot] -= std::min(done_pairs, ev_mark[slot]);
            ev_mark[prev] = 0;
            if (h->h_ctrl[prev].active == 0 && h->h_ctrl[prev].next_sample == total) done = true;
        }
        slot ^= 1;
        guard += ITERS_PER_POLL;
        if (guard > max_iters) {
            (void)hipStreamSynchronize(h->stream);
            return fail(h, "rt_render: iteration bound exceeded (paths did not finish)");
        }
    }
    RT_HIP(hipEventRecord(h->ev_end, h->stream));
    RT_HIP(hipStreamSynchronize(h->stream));
    // the batch enqueued last holds the final counters
    Ctrl fin;
    RT_HIP(hipMemcpy(&fin, h->d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
    h->stats.segments = fin.segments;
    h->stats.camera_rays = fin.camera_rays;
    h->stats.dropped = fin.dropped;
    h->stats.iterations = fin.iterations;
    float ms = 0.f;
    RT_HIP(hipEventElapsedTime(&ms, h->ev_begin, h->ev_end));
    h->stats.render_ms = ms;
    drain_timers(h);
    return 0;
}

int rt_intersect_batch(rt_handle *h, uint32_t n, const float *o, const float *d, const float *time,
                       const int32_t *origin_prim, float tmin, float tmax, int32_t *hit, int32_t *prim, float *t, float *u,
                       float *v) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (!h->have_scene) return fail(h, "rt_intersect_batch: no scene uploaded");
    if (n == 0) return 0;
    if (!o || !d || !hit || !prim || !t || !u || !v) return fail(h, "rt_intersect_batch: null array");
    std::vector<uint32_t> from;
    if (origin_prim) {
        from.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            int32_t id = origin_prim[i];
            if (id >= int32_t(h->prim_to_ref.size())) return fail(h, "rt_intersect_batch: origin_prim out of range");
            from[i] = id < 0 ? RT_REF_NONE : h->prim_to_ref[id];
        }
    }
    // one staging buffer: o[3n] d[3n] time[n] from[n] | t,u,v[3n] hit,prim[2n]
    void *stage = nullptr;
    if (ensure_stage(h, size_t(n) * 13 * sizeof(float), &stage)) return -1;
    float *d_o = static_cast<float *>(stage), *d_d = d_o + 3 * size_t(n), *d_time = time ? d_d + 3 * size_t(n) : nullptr;
    uint32_t *d_from = origin_prim ? reinterpret_cast<uint32_t *>(d_o + 7 * size_t(n)) : nullptr;
    float *d_f = d_o + 8 * size_t(n);
    int *d_i = reinterpret_cast<int *>(d_o + 11 * size_t(n));
    hipError_t e = hipSuccess;
    if (origin_prim) e = hipMemcpyAsync(d_from, from.data(), size_t(n) * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_o, o, size_t(n) * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d, d, size_t(n) * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && time) e = hipMemcpyAsync(d_time, time, size_t(n) * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        dim3 grid((n + BLOCK - 1) / BLOCK), block(BLOCK);
        const size_t dyn_lds = h->dyn_lds4();
        if (h->accel == ACCEL_FLAT)
            k_intersect_batch<ACCEL_FLAT><<<grid, block, 0, h->stream>>>(h->d_sc, n, d_o, d_d, d_time, d_from, tmin, tmax, d_i,
                                                                         d_i + n, d_f, d_f + n, d_f + 2 * size_t(n));
        else if (h->accel == ACCEL_BVH4)
            k_intersect_batch<ACCEL_BVH4><<<grid, block, dyn_lds, h->stream>>>(h->d_sc, n, d_o, d_d, d_time, d_from, tmin, tmax,
                                                                               d_i, d_i + n, d_f, d_f + n, d_f + 2 * size_t(n));
        else if (h->accel == ACCEL_BVH4Q)
            k_intersect_batch<ACCEL_BVH4Q><<<grid, block, dyn_lds, h->stream>>>(h->d_sc, n, d_o, d_d, d_time, d_from, tmin, tmax,
                                                                                d_i, d_i + n, d_f, d_f + n, d_f + 2 * size_t(n));
        else
            k_intersect_batch<ACCEL_BVH2><<<grid, block, 0, h->stream>>>(h->d_sc, n, d_o, d_d, d_time, d_from, tmin, tmax, d_i,
                                                                         d_i + n, d_f, d_f + n, d_f + 2 * size_t(n));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hit, d_i, size_t(n) * sizeof(int), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(prim, d_i + n, size_t(n) * sizeof(int), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t, d_f, size_t(n) * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(u, d_f + n, size_t(n) * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(v, d_f + 2 * size_t(n), size_t(n) * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, std::string("rt_intersect_batch: ") + hipGetErrorString(e));
    return 0;
}

int rt_camera_rays(rt_handle *h, const rt_camera *cam, const rt_render_params *params, uint32_t n, const uint32_t *pixel,
                   const uint32_t *stratum, float *o, float *d, float *time) {
    if (!h) return -1;
    rt_render_params own;
    if (read_params(h, params, &own)) return -1;
    const rt_render_params *p = &own;
    RT_HIP(hipSetDevice(h->device));
    DCamera dc;
    uint32_t se;
    if (make_camera(h, cam, p, &dc, &se)) return -1;
    if (n == 0) return 0;
    if (!pixel || !stratum || !o || !d || !time) return fail(h, "rt_camera_rays: null array");
    for (uint32_t i = 0; i < n; ++i)
        if (pixel[i] >= dc.npix || stratum[i] >= dc.sqrt_spp * dc.sqrt_spp)
            return fail(h, "rt_camera_rays: pixel or stratum out of range");
    void *stage = nullptr;
    if (ensure_stage(h, size_t(n) * 9 * sizeof(float), &stage)) return -1;
    uint32_t *d_u = static_cast<uint32_t *>(stage);
    float *d_f = reinterpret_cast<float *>(d_u + 2 * size_t(n));
    hipError_t e = hipMemcpyAsync(d_u, pixel, size_t(n) * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_u + n, stratum, size_t(n) * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        k_camera_rays<<<(n + 255) / 256, 256, 0, h->stream>>>(h->d_cam, uint32_t(p->seed), uint32_t(p->seed >> 32), n, d_u, d_u + n,
                                                              d_f, d_f + 3 * size_t(n), d_f + 6 * size_t(n));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(o, d_f, size_t(n) * 12, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d, d_f + 3 * size_t(n), size_t(n) * 12, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(time, d_f + 6 * size_t(n), size_t(n) * 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, std::string("rt_camera_rays: ") + hipGetErrorString(e));
    return 0;
}

int rt_eval_texture(rt_handle *h, uint32_t tex, uint32_t n, const float *uvp, float *rgb) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (!h->have_scene) return fail(h, "rt_eval_texture: no scene uploaded");
    if (tex >= h->n_textures) return fail(h, "rt_eval_texture: texture id out of range");
    if (n == 0) return 0;
    if (!uvp || !rgb) return fail(h, "rt_eval_texture: null array");
    void *stage = nullptr;
    if (ensure_stage(h, size_t(n) * 8 * sizeof(float), &stage)) return -1;
    float *d_in = static_cast<float *>(stage), *d_out = d_in + 5 * size_t(n);
    hipError_t e = hipMemcpyAsync(d_in, uvp, size_t(n) * 20, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        k_eval_texture<<<(n + 255) / 256, 256, 0, h->stream>>>(h->d_sc, tex, n, d_in, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rgb, d_out, size_t(n) * 12, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, std::string("rt_eval_texture: ") + hipGetErrorString(e));
    return 0;
}

int rt_philox(rt_handle *h, uint32_t n, const uint32_t *ctr, const uint32_t key[2], uint32_t *out) {
    if (!h) return -1;
    RT_HIP(hipSetDevice(h->device));
    if (n == 0) return 0;
    if (!ctr || !key || !out) return fail(h, "rt_philox: null array");
    void *stage = nullptr;
    if (ensure_stage(h, size_t(n) * 32, &stage)) return -1;
    uint32_t *d_in = static_cast<uint32_t *>(stage), *d_out = d_in + 4 * size_t(n);
    hipError_t e = hipMemcpyAsync(d_in, ctr, size_t(n) * 16, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        k_philox<<<(n + 255) / 256, 256, 0, h->stream>>>(n, d_in, key[0], key[1], d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, size_t(n) * 16, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, std::string("rt_philox: ") + hipGetErrorString(e));
    return 0;
}

} // extern "C"
