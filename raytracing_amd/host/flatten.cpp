// flatten.cpp -- scene graph -> flat fp32 arrays (rt_scene_desc).
//
// * Translate / RotateY (hittable.rs:679-818) are rigid maps; quads and spheres are closed
//   under them and the reference never normalises ray directions, so t is unchanged when the
//   map is applied to the primitive instead of the ray.  They are baked here in f64.
// * HittableVec keeps its order (hittable.rs:252-264 resolves ties by visiting order); a list
//   nested directly inside a list is spliced in place, which visits the same primitives in
//   the same order with the same shrinking interval.
// * BVHNode keeps the topology BVHNode::new built (boundingbox.rs:179-206); child boxes are
//   recomputed from the baked primitives and rounded outward to fp32.
#include "raytracing.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <queue>

namespace raytracing {
namespace {

constexpr double INF = std::numeric_limits<double>::infinity();

struct Xf { // object -> world: p' = M p + T
    double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double t[3] = {0, 0, 0};
    bool identity = true;
    Vec3 point(const Vec3 &p) const {
        return {m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + t[0], m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + t[1],
                m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + t[2]};
    }
    Vec3 vec(const Vec3 &p) const {
        return {m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z, m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z,
                m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z};
    }
    // this o inner
    Xf compose(const Xf &in) const {
        Xf r;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][0] * in.m[0][j] + m[i][1] * in.m[1][j] + m[i][2] * in.m[2][j];
            r.t[i] = m[i][0] * in.t[0] + m[i][1] * in.t[1] + m[i][2] * in.t[2] + t[i];
        }
        r.identity = false;
        return r;
    }
};

struct Box {
    double lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
    void add(const Vec3 &p) {
        for (int i = 0; i < 3; ++i) lo[i] = std::fmin(lo[i], p[i]), hi[i] = std::fmax(hi[i], p[i]);
    }
    void add(const Box &b) {
        for (int i = 0; i < 3; ++i) lo[i] = std::fmin(lo[i], b.lo[i]), hi[i] = std::fmax(hi[i], b.hi[i]);
    }
};

float round_down(double x) {
    float f = float(x);
    return double(f) > x ? std::nextafterf(f, -std::numeric_limits<float>::infinity()) : f;
}
float round_up(double x) {
    float f = float(x);
    return double(f) < x ? std::nextafterf(f, std::numeric_limits<float>::infinity()) : f;
}

struct Flattener {
    FlatScene &fs;
    std::map<const Material *, uint32_t> mat_ids;
    std::map<const Texture *, uint32_t> tex_ids;
    std::map<const Hittable *, uint32_t> identity_refs; // leaves flattened without a transform
    std::vector<Box> node_boxes;                        // per node: box of everything below it
    int next_auto_id = 0;
    int max_prim_id = -1;

    explicit Flattener(FlatScene &f) : fs(f) {}

    uint32_t texture(const Texture *t) {
        if (!t) throw Error("flatten: null texture");
        auto it = tex_ids.find(t);
        if (it != tex_ids.end()) return it->second;
        rt_texture r{};
        switch (t->kind()) {
        case Texture::SOLID: {
            auto *s = static_cast<const SolidColor *>(t);
            r.type = RT_TEX_SOLID;
            r.color[0] = float(s->albedo.r), r.color[1] = float(s->albedo.g), r.color[2] = float(s->albedo.b);
            break;
        }
        case Texture::CHECKER: {
            auto *c = static_cast<const Checkerboard *>(t);
            r.type = RT_TEX_CHECKER;
            r.a = texture(c->even.get()); // children first: their ids precede the parent's
            r.b = texture(c->odd.get());
            r.scale = float(1.0 / c->scale); // texture.rs:61 multiplies by 1/scale
            break;
        }
        case Texture::IMAGE: {
            auto *im = static_cast<const ImageTexture *>(t);
            r.type = RT_TEX_IMAGE;
            r.a = im->width, r.b = im->height;
            r.offset = uint32_t(fs.texels.size());
            fs.texels.insert(fs.texels.end(), im->image_data.begin(), im->image_data.end());
            break;
        }
        case Texture::NOISE: {
            auto *n = static_cast<const NoiseTexture *>(t);
            r.type = RT_TEX_NOISE;
            r.scale = float(n->scale);
            r.offset = uint32_t(fs.noise_perm.size() / 768);
            for (int i = 0; i < 256; ++i) {
                fs.noise_vec.push_back(float(n->ranvec[i].x));
                fs.noise_vec.push_back(float(n->ranvec[i].y));
                fs.noise_vec.push_back(float(n->ranvec[i].z));
                fs.noise_vec.push_back(0.f);
            }
            for (int a = 0; a < 3; ++a)
                for (int i = 0; i < 256; ++i) fs.noise_perm.push_back(n->perm[a][i]);
            break;
        }
        }
        fs.textures.push_back(r);
        uint32_t id = uint32_t(fs.textures.size() - 1);
        tex_ids[t] = id;
        return id;
    }

    bool texture_needs_uv(const Texture *t) {
        if (t->kind() == Texture::IMAGE) return true;
        if (t->kind() == Texture::CHECKER) {
            auto *c = static_cast<const Checkerboard *>(t);
            return texture_needs_uv(c->even.get()) || texture_needs_uv(c->odd.get());
        }
        return false;
    }

    uint32_t material(const Material *m) {
        if (!m) throw Error("flatten: null material");
        auto it = mat_ids.find(m);
        if (it != mat_ids.end()) return it->second;
        rt_material r{};
        const Texture *tex = nullptr;
        switch (m->kind()) {
        case Material::LAMBERTIAN: r.type = RT_MAT_LAMBERTIAN, tex = static_cast<const Lambertian *>(m)->texture.get(); break;
        case Material::LIGHT: r.type = RT_MAT_LIGHT, tex = static_cast<const DiffuseLight *>(m)->texture.get(); break;
        case Material::ISOTROPIC: r.type = RT_MAT_ISOTROPIC, tex = static_cast<const Isotropic *>(m)->texture.get(); break;
        case Material::METAL: {
            auto *mm = static_cast<const Metal *>(m);
            r.type = RT_MAT_METAL;
            r.albedo[0] = float(mm->albedo.r), r.albedo[1] = float(mm->albedo.g), r.albedo[2] = float(mm->albedo.b);
            r.param = float(mm->fuzz);
            break;
        }
        case Material::DIELECTRIC:
            r.type = RT_MAT_DIELECTRIC;
            r.param = float(static_cast<const Dielectric *>(m)->refraction_index);
            break;
        }
        if (tex) {
            r.tex = texture(tex);
            r.needs_uv = texture_needs_uv(tex) ? 1u : 0u;
        }
        fs.materials.push_back(r);
        uint32_t id = uint32_t(fs.materials.size() - 1);
        mat_ids[m] = id;
        return id;
    }

    int prim_id_of(const Hittable &h) {
        int id = h.prim_id >= 0 ? h.prim_id : next_auto_id;
        next_auto_id = std::max(next_auto_id, id + 1);
        max_prim_id = std::max(max_prim_id, id);
        return id;
    }

    uint32_t quad(const Parallelogram &q, const Xf &xf, Box &box) {
        Vec3 Q = xf.point(q.corner), u = xf.vec(q.u), v = xf.vec(q.v);
        // Parallelogram::new (hittable.rs:295-319) on the mapped corner and edges
        Vec3 n = Vec3::cross(u, v);
        double nn = Vec3::dot(n, n);
        if (!(nn > 0.0) || !std::isfinite(nn)) throw Error("flatten: degenerate quad (u x v = 0)");
        Vec3 normal = n.as_unit();
        double d = Vec3::dot(normal, Q);
        Vec3 w = n / nn;
        double area = n.len();
        Vec3 avec = Vec3::cross(v, w), bvec = Vec3::cross(w, u);
        rt_quad r{};
        auto put = [](float *dst, const Vec3 &s) { dst[0] = float(s.x), dst[1] = float(s.y), dst[2] = float(s.z); };
        put(r.q, Q), put(r.n, normal), put(r.avec, avec), put(r.bvec, bvec), put(r.u, u), put(r.v, v);
        r.d = float(d);
        r.area = float(area);
        r.shape = q.shape;
        r.mat = material(q.material.get());
        r.prim_id = uint32_t(prim_id_of(q));
        fs.quads.push_back(r);
        for (const Vec3 &p : {Q, Q + u, Q + v, Q + u + v}) box.add(p);
        return (RT_KIND_QUAD << RT_REF_KIND_SHIFT) | uint32_t(fs.quads.size() - 1);
    }

    uint32_t sphere(const Sphere &s, const Xf &xf, Box &box) {
        if (!(s.radius > 0.0)) throw Error("flatten: sphere radius must be > 0");
        Vec3 c0 = xf.point(s.center0), cv = xf.vec(s.center_vec);
        rt_sphere r{};
        r.c0[0] = float(c0.x), r.c0[1] = float(c0.y), r.c0[2] = float(c0.z);
        r.cv[0] = float(cv.x), r.cv[1] = float(cv.y), r.cv[2] = float(cv.z);
        r.radius = float(s.radius);
        r.mat = material(s.material.get());
        r.prim_id = uint32_t(prim_id_of(s));
        r.moving = (cv.x != 0.0 || cv.y != 0.0 || cv.z != 0.0) ? 1u : 0u;
        // net rotation about y of the baked map: m[0][0] = cos, m[0][2] = sin
        r.u_shift = float(-std::atan2(xf.m[0][2], xf.m[0][0]) / (2.0 * 3.14159265358979323846));
        fs.spheres.push_back(r);
        // a little slack for the fp32 copies of centre and radius
        double rad = s.radius * (1.0 + 1e-6) + 1e-6 * (std::fabs(c0.x) + std::fabs(c0.y) + std::fabs(c0.z));
        Vec3 rv{rad, rad, rad};
        box.add(c0 - rv), box.add(c0 + rv), box.add(c0 + cv - rv), box.add(c0 + cv + rv);
        return (RT_KIND_SPHERE << RT_REF_KIND_SHIFT) | uint32_t(fs.spheres.size() - 1);
    }

    // entries of a list under xf, nested lists spliced in order
    void list_entries(const HittableVec &l, const Xf &xf, std::vector<uint32_t> &out, Box &box) {
        for (const auto &o : l.objects) {
            // peel transforms to see whether a list sits underneath
            const Hittable *h = o.get();
            Xf x = xf;
            while (h->kind() == Hittable::TRANSLATE || h->kind() == Hittable::ROTATE_Y) h = unwrap(h, x);
            if (h->kind() == Hittable::LIST) {
                list_entries(*static_cast<const HittableVec *>(h), x, out, box);
            } else {
                Box b;
                out.push_back(object(*h, x, b));
                box.add(b);
            }
        }
    }

    const Hittable *unwrap(const Hittable *h, Xf &x) {
        if (h->kind() == Hittable::TRANSLATE) {
            auto *t = static_cast<const Translate *>(h);
            Xf m;
            m.t[0] = t->offset.x, m.t[1] = t->offset.y, m.t[2] = t->offset.z;
            x = x.compose(m);
            return t->object.get();
        }
        auto *r = static_cast<const RotateY *>(h);
        Xf m; // object -> world of RotateY (hittable.rs:795-805)
        m.m[0][0] = r->cos_theta, m.m[0][2] = r->sin_theta;
        m.m[2][0] = -r->sin_theta, m.m[2][2] = r->cos_theta;
        x = x.compose(m);
        return r->object.get();
    }

    uint32_t leaf_node(const std::vector<uint32_t> &entries, const Box &box) {
        rt_node n{};
        n.left = uint32_t(fs.refs.size());
        n.right = RT_NODE_LEAF;
        n.count = uint32_t(entries.size());
        fs.refs.insert(fs.refs.end(), entries.begin(), entries.end());
        fs.nodes.push_back(n);
        node_boxes.push_back(box);
        return (RT_KIND_NODE << RT_REF_KIND_SHIFT) | uint32_t(fs.nodes.size() - 1);
    }

    uint32_t object(const Hittable &h, const Xf &xf, Box &box) {
        switch (h.kind()) {
        case Hittable::SPHERE: {
            uint32_t r = sphere(static_cast<const Sphere &>(h), xf, box);
            if (xf.identity) identity_refs[&h] = r;
            return r;
        }
        case Hittable::QUAD: {
            uint32_t r = quad(static_cast<const Parallelogram &>(h), xf, box);
            if (xf.identity) identity_refs[&h] = r;
            return r;
        }
        case Hittable::TRANSLATE:
        case Hittable::ROTATE_Y: {
            Xf x = xf;
            const Hittable *in = unwrap(&h, x);
            return object(*in, x, box);
        }
        case Hittable::LIST: {
            std::vector<uint32_t> entries;
            list_entries(static_cast<const HittableVec &>(h), xf, entries, box);
            return leaf_node(entries, box);
        }
        case Hittable::MEDIUM: {
            auto &m = static_cast<const ConstantMedium &>(h);
            std::vector<uint32_t> entries;
            Box b;
            const Hittable *bh = m.boundary.get();
            Xf x = xf;
            while (bh->kind() == Hittable::TRANSLATE || bh->kind() == Hittable::ROTATE_Y) bh = unwrap(bh, x);
            if (bh->kind() == Hittable::LIST) {
                list_entries(*static_cast<const HittableVec *>(bh), x, entries, b);
            } else if (bh->kind() == Hittable::SPHERE || bh->kind() == Hittable::QUAD) {
                entries.push_back(object(*bh, x, b));
            } else {
                throw Error("flatten: ConstantMedium boundary must be a sphere, a quad or a list of them");
            }
            for (uint32_t e : entries) {
                uint32_t k = e >> RT_REF_KIND_SHIFT;
                if (k != RT_KIND_QUAD && k != RT_KIND_SPHERE)
                    throw Error("flatten: ConstantMedium boundary must be a sphere, a quad or a list of them");
            }
            if (entries.empty()) throw Error("flatten: ConstantMedium with an empty boundary");
            rt_medium r{};
            r.first = uint32_t(fs.refs.size());
            r.count = uint32_t(entries.size());
            fs.refs.insert(fs.refs.end(), entries.begin(), entries.end());
            r.neg_inv_density = float(m.neg_inv_density);
            r.mat = material(m.phase_fn.get());
            r.prim_id = uint32_t(prim_id_of(h));
            fs.media.push_back(r);
            box.add(b);
            return (RT_KIND_MEDIUM << RT_REF_KIND_SHIFT) | uint32_t(fs.media.size() - 1);
        }
        case Hittable::BVH: {
            auto &n = static_cast<const BVHNode &>(h);
            if (n.left == n.right) return object(*n.left, xf, box); // 1-object node: same hit, tested once
            uint32_t idx = uint32_t(fs.nodes.size());
            fs.nodes.push_back(rt_node{});
            node_boxes.push_back(Box{});
            Box bl, br;
            uint32_t l = object(*n.left, xf, bl);
            uint32_t r = object(*n.right, xf, br);
            rt_node &nd = fs.nodes[idx];
            nd.left = l, nd.right = r;
            pad_store(bl, nd.lmin, nd.lmax);
            pad_store(br, nd.rmin, nd.rmax);
            box.add(bl), box.add(br);
            node_boxes[idx] = box;
            return (RT_KIND_NODE << RT_REF_KIND_SHIFT) | idx;
        }
        }
        throw Error("flatten: unknown hittable kind");
    }

    // BoundingBox3::new pads thin axes to 1e-4 (boundingbox.rs:14-30); then round outward
    static void pad_store(const Box &b, float *mn, float *mx) {
        for (int i = 0; i < 3; ++i) {
            double lo = b.lo[i], hi = b.hi[i];
            if (hi - lo < 0.0001) lo -= 0.00005, hi += 0.00005;
            mn[i] = round_down(lo);
            mx[i] = round_up(hi);
        }
    }

    // stack entries traverse() may hold while below `ref`
    uint32_t stack_need(uint32_t ref) {
        if ((ref >> RT_REF_KIND_SHIFT) != RT_KIND_NODE) return 0;
        const rt_node &n = fs.nodes[ref & RT_REF_INDEX_MASK];
        if (n.right == RT_NODE_LEAF) {
            uint32_t k = 0, deepest = 0;
            for (uint32_t i = 0; i < n.count; ++i) {
                uint32_t e = fs.refs[n.left + i];
                if ((e >> RT_REF_KIND_SHIFT) == RT_KIND_NODE) ++k, deepest = std::max(deepest, stack_need(e));
            }
            return k + deepest;
        }
        return 1 + std::max(stack_need(n.left), stack_need(n.right));
    }

    // breadth-first renumbering so the top of the tree has the smallest node indices
    void renumber(uint32_t &root) {
        const uint32_t N = uint32_t(fs.nodes.size());
        if (N == 0) return;
        std::vector<uint32_t> order, newidx(N, UINT32_MAX);
        std::queue<uint32_t> qu;
        auto visit = [&](uint32_t ref) {
            if ((ref >> RT_REF_KIND_SHIFT) != RT_KIND_NODE) return;
            uint32_t i = ref & RT_REF_INDEX_MASK;
            if (newidx[i] != UINT32_MAX) return;
            newidx[i] = uint32_t(order.size());
            order.push_back(i);
            qu.push(i);
        };
        visit(root);
        while (!qu.empty()) {
            uint32_t i = qu.front();
            qu.pop();
            const rt_node &n = fs.nodes[i];
            if (n.right == RT_NODE_LEAF) {
                for (uint32_t k = 0; k < n.count; ++k) visit(fs.refs[n.left + k]);
            } else {
                visit(n.left);
                visit(n.right);
            }
        }
        // nodes not reachable from the root keep a slot at the end
        for (uint32_t i = 0; i < N; ++i)
            if (newidx[i] == UINT32_MAX) newidx[i] = uint32_t(order.size()), order.push_back(i);
        auto remap = [&](uint32_t ref) {
            if ((ref >> RT_REF_KIND_SHIFT) != RT_KIND_NODE) return ref;
            return (RT_KIND_NODE << RT_REF_KIND_SHIFT) | newidx[ref & RT_REF_INDEX_MASK];
        };
        std::vector<rt_node> nn(N);
        for (uint32_t k = 0; k < N; ++k) {
            rt_node n = fs.nodes[order[k]];
            if (n.right != RT_NODE_LEAF) n.left = remap(n.left), n.right = remap(n.right);
            nn[k] = n;
        }
        fs.nodes.swap(nn);
        for (auto &r : fs.refs) r = remap(r);
        root = remap(root);
    }

    uint32_t light_prim(const Hittable &h) {
        if (h.kind() == Hittable::QUAD) {
            if (static_cast<const Parallelogram &>(h).shape != RT_SHAPE_PARALLELOGRAM)
                throw Error("flatten: only Parallelogram implements pdf_value/random (hittable.rs:377-395)");
        } else if (h.kind() == Hittable::SPHERE) {
            auto &s = static_cast<const Sphere &>(h);
            if (s.center_vec.x != 0 || s.center_vec.y != 0 || s.center_vec.z != 0)
                throw Error("flatten: a moving sphere cannot be a light");
        } else {
            throw Error("flatten: lights must be a Parallelogram, a Sphere or a HittableVec of those");
        }
        auto it = identity_refs.find(&h);
        if (it != identity_refs.end()) return it->second; // the world's own primitive, not a copy
        Box b;
        Xf id;
        if (h.kind() == Hittable::QUAD) return quad(static_cast<const Parallelogram &>(h), id, b);
        return sphere(static_cast<const Sphere &>(h), id, b);
    }
};

} // namespace

void FlatScene::finish() {
    desc.n_nodes = uint32_t(nodes.size()), desc.nodes = nodes.data();
    desc.n_refs = uint32_t(refs.size()), desc.refs = refs.data();
    desc.n_quads = uint32_t(quads.size()), desc.quads = quads.data();
    desc.n_spheres = uint32_t(spheres.size()), desc.spheres = spheres.data();
    desc.n_media = uint32_t(media.size()), desc.media = media.data();
    desc.n_materials = uint32_t(materials.size()), desc.materials = materials.data();
    desc.n_textures = uint32_t(textures.size()), desc.textures = textures.data();
    desc.n_lights = uint32_t(light_refs.size()), desc.light_refs = light_refs.data();
    desc.n_texel_bytes = uint32_t(texels.size()), desc.texels = texels.data();
    desc.n_noise = uint32_t(noise_perm.size() / 768);
    desc.noise_vec = noise_vec.data(), desc.noise_perm = noise_perm.data();
}

std::unique_ptr<FlatScene> flatten(const Hittable &world, const Hittable *lights) {
    auto fs = std::make_unique<FlatScene>();
    Flattener f(*fs);
    Box box;
    Xf id;
    uint32_t root = f.object(world, id, box);
    fs->desc.lights_kind = RT_LIGHTS_NONE;
    if (lights) {
        if (lights->kind() == Hittable::LIST) {
            auto *l = static_cast<const HittableVec *>(lights);
            if (l->objects.empty()) throw Error("flatten: empty light list");
            for (auto &o : l->objects) fs->light_refs.push_back(f.light_prim(*o));
            fs->desc.lights_kind = RT_LIGHTS_LIST;
        } else {
            fs->light_refs.push_back(f.light_prim(*lights));
            fs->desc.lights_kind = RT_LIGHTS_SINGLE;
        }
    }
    if (fs->materials.empty()) throw Error("flatten: scene has no materials");
    if (fs->nodes.size() > RT_REF_INDEX_MASK || fs->quads.size() > RT_REF_INDEX_MASK || fs->spheres.size() > RT_REF_INDEX_MASK)
        throw Error("flatten: scene too large for 29-bit references");
    if (fs->texels.size() > 0xFFFFFFFFull) throw Error("flatten: more than 4 GiB of texels");
    f.renumber(root);
    fs->desc.root = root;
    fs->desc.max_stack = f.stack_need(root);
    fs->desc.n_prims = uint32_t(f.max_prim_id + 1);
    fs->finish();
    return fs;
}

} // namespace raytracing
