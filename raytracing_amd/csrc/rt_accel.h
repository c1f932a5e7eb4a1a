// rt_accel.h -- host side of the 4-wide BVH the device traverses (built at scene upload).
//
// The scene descriptor carries the reference's structure: binary BVHNode nodes in the topology
// BVHNode::new built (boundingbox.rs:179-206) and HittableVec lists (hittable.rs:252-264).
// The closest hit does not depend on how the primitives are grouped, so the upload derives a
// structure that suits the hardware better and leaves the descriptor as it is:
//   * every box is recomputed from the fp32 primitive records the kernels test, rounded outward;
//   * a list becomes a small tree over its entries (median splits), so lists are culled by boxes
//     like everything else and the device has one node type;
//   * binary nodes are collapsed into 4-wide nodes: the child with the largest box is replaced by
//     its own children until four slots are used -- half the dependent node fetches per ray;
//   * optionally (RT_BUILD_SAH) the binary tree is rebuilt over all primitives with a binned
//     surface-area heuristic before the collapse.
// Nodes are numbered breadth first.
#pragma once
#include "../../include/rt_amd.h"
#include "rt_flat.h"
#include "rt_leafword.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <queue>
#include <string>
#include <vector>

namespace rt {

// 128 bytes: child boxes as structure of arrays, then the child references.  An unused slot has
// ref RT_REF_NONE and an inverted box (+inf .. -inf) that no ray enters.
struct Node4 {
    float lo[3][4];
    float hi[3][4];
    uint32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(Node4) == 128, "wide node is one 128-byte line");

// The same node in 64 bytes: child boxes as bytes on a grid of the node's own.  Per axis
// plane = origin + q * 2^(exp - 127); lower planes are rounded down and upper planes up, after
// widening the box by a few fp32 ulps of the node's coordinates (the device evaluates
// q * (s / d) + (origin - o) / d in fp32), so a quantised box always contains the fp32 one.
struct Node4Q {
    float origin[3];
    uint32_t exps;      // biased exponents of the grid steps: x | y << 8 | z << 16
    uint32_t lo[3];     // one byte per child
    uint32_t hi[3];
    uint32_t spare[2];
    uint32_t child[4];
};
static_assert(sizeof(Node4Q) == 64, "quantised wide node is half a 128-byte line");

// Six quads of one list that are the faces of a box along the world's axes, as one leaf of the wide
// tree (RT_KIND_BOX): 48 bytes.  The planes are the coordinates the faces' own quad records hold, so the
// slab test's t = (plane - o) / d is the quad test's own quotient for a unit normal along an axis, bit for
// bit (ties between faces of neighbouring boxes fall as they do between the quads).
struct WideBox {
    float lo[3];
    uint32_t ref_lo_x;
    float hi[3];
    uint32_t ref_hi_x;
    uint32_t ref_lo_y, ref_lo_z, ref_hi_y, ref_hi_z;
};
static_assert(sizeof(WideBox) == 48, "three 16-byte loads per box");

namespace accel {

struct Box {
    float lo[3], hi[3];
    Box() {
        for (int i = 0; i < 3; ++i) lo[i] = std::numeric_limits<float>::infinity(), hi[i] = -lo[i];
    }
    void add(const Box &b) {
        for (int i = 0; i < 3; ++i) lo[i] = std::min(lo[i], b.lo[i]), hi[i] = std::max(hi[i], b.hi[i]);
    }
    void add(const double p[3]) {
        for (int i = 0; i < 3; ++i) {
            // outward to fp32, plus one ulp: the device evaluates hit points in fp32
            float d = std::nextafter(float(p[i]), -std::numeric_limits<float>::infinity());
            float u = std::nextafter(float(p[i]), std::numeric_limits<float>::infinity());
            lo[i] = std::min(lo[i], d), hi[i] = std::max(hi[i], u);
        }
    }
    bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
    double area() const {
        if (!valid()) return 0.0;
        double x = double(hi[0]) - lo[0], y = double(hi[1]) - lo[1], z = double(hi[2]) - lo[2];
        return 2.0 * (x * y + y * z + z * x);
    }
    double centre(int a) const { return 0.5 * (double(lo[a]) + double(hi[a])); }
    // BoundingBox3::new pads thin axes to 1e-4 (boundingbox.rs:14-30)
    void pad() {
        for (int i = 0; i < 3; ++i)
            if (double(hi[i]) - double(lo[i]) < 1e-4) {
                double c = centre(i);
                lo[i] = std::nextafter(float(c - 5e-5), -std::numeric_limits<float>::infinity());
                hi[i] = std::nextafter(float(c + 5e-5), std::numeric_limits<float>::infinity());
            }
    }
};

// A child slot of the intermediate tree: a primitive/medium ref, or an inner node (index into `inner`).
struct Slot {
    uint32_t ref;   // primitive ref, or RT_REF_NONE when `node` is used
    int32_t node;   // index into Builder::inner, -1 for a primitive
    Box box;
};
struct Inner {
    int32_t a, b; // indices into Builder::slots
};

struct Stats {
    uint32_t nodes = 0, depth = 0, stack_need = 0, leaves = 0;
    // surface-area cost: expected boxes tested by a random ray that enters the root box =
    // 4 * sum over nodes of area(node) / area(root)
    double cost = 0.0;
    // measured: mean work of walking a surface-to-surface segment through the tree (walk_cost), in node visits
    double walk = 0.0;
};

struct Builder {
    const rt_scene_desc *s;
    std::vector<Slot> slots;
    std::vector<Inner> inner;
    std::vector<int32_t> node_slot; // descriptor node index -> slot (memo: nodes may be shared)
    std::string error;
    bool find_boxes = true;      // runs of six quads in a list that are an axis-aligned box become one leaf
    std::vector<WideBox> boxes;  // the records of those leaves
    std::vector<uint32_t> box_faces; // 6 quad refs per box, list order (for checks)

    explicit Builder(const rt_scene_desc *scene) : s(scene), node_slot(scene->n_nodes, -1) {}

    uint32_t class_of(uint32_t ref) const {
        const uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
        const uint32_t mat = kind == RT_KIND_QUAD ? s->quads[idx].mat : kind == RT_KIND_SPHERE ? s->spheres[idx].mat : s->media[idx].mat;
        const uint32_t type = s->materials[mat].type;
        return type == RT_MAT_LIGHT ? RT4_CLASS_END : type == RT_MAT_METAL ? RT4_CLASS_METAL : type == RT_MAT_DIELECTRIC ? RT4_CLASS_GLASS : RT4_CLASS_DIFFUSE;
    }

    // refs[0..6) of a list: the faces of one box along the world's axes, all of one material class?
    // Then the box's record is appended and its leaf reference returned, else RT_REF_NONE.
    uint32_t match_wide_box(const uint32_t *refs) {
        const rt_quad *faces[6];
        for (int i = 0; i < 6; ++i) {
            if ((refs[i] >> RT_REF_KIND_SHIFT) != RT_KIND_QUAD) return RT_REF_NONE;
            faces[i] = &s->quads[refs[i] & RT_REF_INDEX_MASK];
            if (class_of(refs[i]) != class_of(refs[0])) return RT_REF_NONE;
        }
        float rec[32];
        if (!flat::match_box(faces, refs, 6, rec)) return RT_REF_NONE;
        uint32_t cls;
        std::memcpy(&cls, &rec[24], 4);
        if (cls != RT_BOX_ALIGNED) return RT_REF_NONE;
        if (boxes.size() >= (1u << RT4_CLASS_SHIFT)) return RT_REF_NONE;
        WideBox b;
        uint32_t r[6]; // lo x y z, hi x y z
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = rec[12 + 2 * a], b.hi[a] = rec[13 + 2 * a];
            std::memcpy(&r[a], &rec[18 + a], 4);
            std::memcpy(&r[3 + a], &rec[21 + a], 4);
            if (!(b.lo[a] < b.hi[a])) return RT_REF_NONE;
        }
        b.ref_lo_x = r[0], b.ref_lo_y = r[1], b.ref_lo_z = r[2], b.ref_hi_x = r[3], b.ref_hi_y = r[4], b.ref_hi_z = r[5];
        // match_box lets a face differ from the box by 1e-5 of the scene's scale (enough for a scan that is
        // checked against the quads it replaces at upload of a dozen quads); here every face must span the
        // box to within fp32 rounding of its own corners, so that the slab test and the six interior tests
        // can differ only where those differ among themselves.
        for (int i = 0; i < 6; ++i)
            for (int a = 0; a < 3; ++a) {
                if (std::fabs(faces[i]->n[a]) == 1.0f) continue; // its normal's axis
                double mn = 1e300, mx = -1e300;
                for (int c = 0; c < 4; ++c) {
                    const double x = double(faces[i]->q[a]) + ((c & 1) ? double(faces[i]->u[a]) : 0.0) + ((c & 2) ? double(faces[i]->v[a]) : 0.0);
                    mn = std::min(mn, x), mx = std::max(mx, x);
                }
                const double eps = 4.0 * std::ldexp(std::max(std::fabs(double(b.lo[a])), std::fabs(double(b.hi[a]))), -23);
                if (std::fabs(mn - b.lo[a]) > eps || std::fabs(mx - b.hi[a]) > eps) return RT_REF_NONE;
            }
        boxes.push_back(b);
        box_faces.insert(box_faces.end(), refs, refs + 6);
        return (RT_KIND_BOX << RT_REF_KIND_SHIFT) | uint32_t(boxes.size() - 1);
    }

    // reference -> reference with the class of its material (see RT4_CLASS_SHIFT)
    uint32_t with_class(uint32_t ref) const {
        if ((ref >> RT_REF_KIND_SHIFT) == RT_KIND_BOX) // its faces share one class (match_wide_box)
            return ref | (class_of(box_faces[6 * size_t(ref & RT_REF_INDEX_MASK)]) << RT4_CLASS_SHIFT);
        return ref | (class_of(ref) << RT4_CLASS_SHIFT);
    }

    Box prim_box(uint32_t ref) const {
        const uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
        Box b;
        if (kind == RT_KIND_QUAD) { // the parallelogram Q, Q+u, Q+v, Q+u+v bounds triangle and disc too
            const rt_quad &q = s->quads[idx];
            for (int k = 0; k < 4; ++k) {
                double p[3];
                for (int i = 0; i < 3; ++i) p[i] = double(q.q[i]) + ((k & 1) ? double(q.u[i]) : 0.0) + ((k & 2) ? double(q.v[i]) : 0.0);
                b.add(p);
            }
            b.pad();
        } else if (kind == RT_KIND_SPHERE) { // both ends of the motion (hittable.rs:136-147)
            const rt_sphere &sp = s->spheres[idx];
            const double r = double(sp.radius) * (1.0 + 1e-6);
            for (int e = 0; e < 2; ++e)
                for (int sgn = -1; sgn <= 1; sgn += 2) {
                    double p[3];
                    for (int i = 0; i < 3; ++i) p[i] = double(sp.c0[i]) + e * double(sp.cv[i]) + sgn * r;
                    b.add(p);
                }
        } else if (kind == RT_KIND_MEDIUM) {
            const rt_medium &m = s->media[idx];
            for (uint32_t k = 0; k < m.count; ++k) b.add(prim_box(s->refs[m.first + k]));
        } else if (kind == RT_KIND_BOX) { // what its six faces' boxes cover
            for (int k = 0; k < 6; ++k) b.add(prim_box(box_faces[6 * size_t(idx) + k]));
        }
        return b;
    }

    int32_t new_slot(uint32_t ref, int32_t node, const Box &box) {
        slots.push_back(Slot{ref, node, box});
        return int32_t(slots.size() - 1);
    }
    int32_t join(int32_t a, int32_t b) {
        Box box = slots[a].box;
        box.add(slots[b].box);
        inner.push_back(Inner{a, b});
        return new_slot(RT_REF_NONE, int32_t(inner.size() - 1), box);
    }

    // a binary tree over `items` (slot indices): median split of the centres on the longest axis
    int32_t median_tree(std::vector<int32_t> &items, size_t lo, size_t hi) {
        if (hi - lo == 1) return items[lo];
        Box cb;
        for (size_t i = lo; i < hi; ++i) {
            double c[3] = {slots[items[i]].box.centre(0), slots[items[i]].box.centre(1), slots[items[i]].box.centre(2)};
            cb.add(c);
        }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (double(cb.hi[a]) - cb.lo[a] > double(cb.hi[axis]) - cb.lo[axis]) axis = a;
        size_t mid = lo + (hi - lo) / 2;
        std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi,
                         [&](int32_t x, int32_t y) { return slots[x].box.centre(axis) < slots[y].box.centre(axis); });
        int32_t l = median_tree(items, lo, mid), r = median_tree(items, mid, hi);
        return join(l, r);
    }

    // binned surface-area heuristic, 16 bins, leaves of one primitive (the device has no leaf lists)
    int32_t sah_tree(std::vector<int32_t> &items, size_t lo, size_t hi) {
        const size_t n = hi - lo;
        if (n == 1) return items[lo];
        if (n <= 4) return median_tree(items, lo, hi);
        Box cb;
        for (size_t i = lo; i < hi; ++i) {
            double c[3] = {slots[items[i]].box.centre(0), slots[items[i]].box.centre(1), slots[items[i]].box.centre(2)};
            cb.add(c);
        }
        constexpr int BINS = 16;
        double best = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_bin = 0;
        for (int axis = 0; axis < 3; ++axis) {
            const double c0 = cb.lo[axis], ext = double(cb.hi[axis]) - c0;
            if (!(ext > 0.0)) continue;
            Box bb[BINS];
            size_t cnt[BINS] = {};
            for (size_t i = lo; i < hi; ++i) {
                int k = std::min(BINS - 1, int((slots[items[i]].box.centre(axis) - c0) / ext * BINS));
                k = std::max(k, 0);
                bb[k].add(slots[items[i]].box);
                ++cnt[k];
            }
            double right_area[BINS];
            size_t right_cnt[BINS];
            Box acc;
            size_t c = 0;
            for (int k = BINS - 1; k > 0; --k) {
                acc.add(bb[k]);
                c += cnt[k];
                right_area[k] = acc.area(), right_cnt[k] = c;
            }
            acc = Box();
            c = 0;
            for (int k = 0; k < BINS - 1; ++k) {
                acc.add(bb[k]);
                c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                double cost = acc.area() * double(c) + right_area[k + 1] * double(right_cnt[k + 1]);
                if (cost < best) best = cost, best_axis = axis, best_bin = k;
            }
        }
        if (best_axis < 0) return median_tree(items, lo, hi); // all centres coincide
        const double c0 = cb.lo[best_axis], ext = double(cb.hi[best_axis]) - c0;
        auto mid_it = std::partition(items.begin() + lo, items.begin() + hi, [&](int32_t x) {
            int k = std::min(BINS - 1, int((slots[x].box.centre(best_axis) - c0) / ext * BINS));
            return std::max(k, 0) <= best_bin;
        });
        size_t mid = size_t(mid_it - items.begin());
        if (mid == lo || mid == hi) return median_tree(items, lo, hi);
        int32_t l = sah_tree(items, lo, mid), r = sah_tree(items, mid, hi);
        return join(l, r);
    }

    // descriptor reference -> slot of the intermediate binary tree (reference topology)
    int32_t from_ref(uint32_t ref, uint32_t depth) {
        if ((ref >> RT_REF_KIND_SHIFT) != RT_KIND_NODE) return new_slot(ref, -1, prim_box(ref));
        const uint32_t idx = ref & RT_REF_INDEX_MASK;
        if (node_slot[idx] >= 0) return node_slot[idx];
        if (depth > 4096) {
            error = "scene graph deeper than 4096 levels";
            return -1;
        }
        const rt_node &n = s->nodes[idx];
        int32_t out;
        if (n.right == RT_NODE_LEAF) {
            if (n.count == 0) {
                error = "empty list in the world";
                return -1;
            }
            std::vector<int32_t> items;
            for (uint32_t k = 0; k < n.count; ++k) {
                if (find_boxes && k + 6 <= n.count) {
                    const uint32_t box = match_wide_box(s->refs + n.left + k);
                    if (box != RT_REF_NONE) {
                        items.push_back(new_slot(box, -1, prim_box(box)));
                        k += 5;
                        continue;
                    }
                }
                int32_t c = from_ref(s->refs[n.left + k], depth + 1);
                if (c < 0) return -1;
                items.push_back(c);
            }
            out = median_tree(items, 0, items.size());
        } else {
            int32_t l = from_ref(n.left, depth + 1);
            if (l < 0) return -1;
            int32_t r = from_ref(n.right, depth + 1);
            if (r < 0) return -1;
            out = join(l, r);
        }
        node_slot[idx] = out;
        return out;
    }

    // The binary tree without the leaves whose slot is marked in `drop`; boxes above a dropped leaf are
    // made again from what is left.  Returns the new top, or -1 when nothing is left.
    int32_t prune(int32_t slot, const std::vector<char> &drop) {
        if (slots[slot].node < 0) return drop[slot] ? -1 : slot;
        const Inner in = inner[slots[slot].node];
        const int32_t a = prune(in.a, drop), b = prune(in.b, drop);
        if (a == in.a && b == in.b) return slot;
        if (a < 0) return b;
        if (b < 0) return a;
        return join(a, b);
    }

    // Primitives that enclose the rest of the scene -- a fog that fills the world ("The Next Week"'s final
    // scene wraps everything in a ConstantMedium of radius 5000) -- cannot be culled by any box: every ray
    // tests them, at whatever moment its walk through the tree gets there, with the few lanes of the wave
    // that are at a primitive just then.  They leave the tree and are tested once per ray when its walk has
    // ended (DScene::post_ref), with every lane that has finished -- the closest hit does not depend on the
    // order of the tests.  At most two; spheres and media only (a quad's box is flat).
    std::vector<uint32_t> take_enclosing(int32_t &top) {
        std::vector<uint32_t> post;
        if (slots[top].node < 0) return post;
        std::vector<int32_t> prims;
        collect_prims(top, prims);
        Box all;
        for (int32_t p : prims) all.add(slots[p].box);
        std::vector<char> drop(slots.size(), 0);
        for (int32_t p : prims) {
            const uint32_t kind = slots[p].ref >> RT_REF_KIND_SHIFT;
            if ((kind != RT_KIND_SPHERE && kind != RT_KIND_MEDIUM) || post.size() >= 2) continue;
            if (slots[p].box.area() < 0.9 * all.area()) continue;
            Box others;
            for (int32_t o : prims)
                if (o != p && !drop[o]) others.add(slots[o].box);
            if (!others.valid()) continue; // it is all there is
            bool encloses = true;
            for (int a = 0; a < 3; ++a) encloses = encloses && slots[p].box.lo[a] <= others.lo[a] && slots[p].box.hi[a] >= others.hi[a];
            if (!encloses) continue;
            drop[p] = 1;
            post.push_back(with_class(slots[p].ref));
        }
        if (!post.empty()) {
            drop.resize(slots.size(), 0);
            const int32_t t = prune(top, drop);
            if (t < 0) post.clear(); // (cannot happen: `others` was never empty)
            else top = t;
        }
        return post;
    }

    void collect_prims(int32_t slot, std::vector<int32_t> &out) const {
        std::vector<int32_t> todo{slot};
        while (!todo.empty()) {
            int32_t x = todo.back();
            todo.pop_back();
            if (slots[x].node < 0) {
                out.push_back(x);
            } else {
                todo.push_back(inner[slots[x].node].a);
                todo.push_back(inner[slots[x].node].b);
            }
        }
    }

    // Collapse to 4-wide nodes, numbered breadth first.  Returns the root ref.
    uint32_t collapse(int32_t root, std::vector<Node4> &nodes, Stats &st) {
        nodes.clear();
        struct Pending {
            int32_t slot;
            uint32_t depth;
        };
        std::queue<Pending> qu;
        if (slots[root].node < 0) {
            // a world of one primitive: one node holding it
            Node4 n;
            fill_empty(n);
            set_child(n, 0, with_class(slots[root].ref), slots[root].box);
            nodes.push_back(n);
            st.nodes = 1, st.depth = 1, st.stack_need = 1, st.leaves = 1;
            return (RT_KIND_NODE << RT_REF_KIND_SHIFT) | 0u;
        }
        qu.push({root, 1});
        uint32_t next_index = 1; // index the next queued node will get
        std::vector<uint32_t> depth_of;
        while (!qu.empty()) {
            Pending p = qu.front();
            qu.pop();
            int32_t ch[4];
            int n = 0;
            ch[n++] = inner[slots[p.slot].node].a;
            ch[n++] = inner[slots[p.slot].node].b;
            while (n < 4) { // open the inner child with the largest box
                int pick = -1;
                double area = -1.0;
                for (int i = 0; i < n; ++i)
                    if (slots[ch[i]].node >= 0 && slots[ch[i]].box.area() > area) area = slots[ch[i]].box.area(), pick = i;
                if (pick < 0) break;
                const Inner in = inner[slots[ch[pick]].node];
                ch[pick] = in.a;
                ch[n++] = in.b;
            }
            Node4 nd;
            fill_empty(nd);
            for (int i = 0; i < n; ++i) {
                const Slot &c = slots[ch[i]];
                if (c.node < 0) {
                    set_child(nd, i, with_class(c.ref), c.box);
                    ++st.leaves;
                } else {
                    set_child(nd, i, (RT_KIND_NODE << RT_REF_KIND_SHIFT) | next_index, c.box);
                    ++next_index;
                    qu.push({ch[i], p.depth + 1});
                }
            }
            nodes.push_back(nd);
            st.depth = std::max(st.depth, p.depth);
        }
        st.nodes = uint32_t(nodes.size());
        // a visit pushes at most three siblings and goes one level down
        st.stack_need = 3 * st.depth + 1;
        return (RT_KIND_NODE << RT_REF_KIND_SHIFT) | 0u;
    }

    static void fill_empty(Node4 &n) {
        const float inf = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a)
            for (int i = 0; i < 4; ++i) n.lo[a][i] = inf, n.hi[a][i] = -inf;
        for (int i = 0; i < 4; ++i) n.child[i] = RT_REF_NONE, n.pad[i] = 0;
    }
    static void set_child(Node4 &n, int i, uint32_t ref, const Box &b) {
        for (int a = 0; a < 3; ++a) n.lo[a][i] = b.lo[a], n.hi[a][i] = b.hi[a];
        n.child[i] = ref;
    }
};

// Exact stack need of near-first traversal: the most entries that can be on the stack at once is
// the maximum over root-to-leaf paths of the sum of (children - 1) along the path.
inline uint32_t stack_need(const std::vector<Node4> &nodes) {
    std::vector<uint32_t> need(nodes.size(), 0);
    for (size_t i = nodes.size(); i-- > 0;) { // children have larger indices (breadth-first numbering)
        uint32_t kids = 0, deepest = 0;
        for (int k = 0; k < 4; ++k) {
            uint32_t c = nodes[i].child[k];
            if (c == RT_REF_NONE) continue;
            ++kids;
            if ((c >> RT_REF_KIND_SHIFT) == RT_KIND_NODE) deepest = std::max(deepest, need[c & RT_REF_INDEX_MASK]);
        }
        need[i] = (kids ? kids - 1 : 0) + deepest;
    }
    return nodes.empty() ? 0 : need[0] + 1;
}

inline double sa_cost(const std::vector<Node4> &nodes) {
    if (nodes.empty()) return 0.0;
    auto child_area = [&](const Node4 &n, int k) {
        Box b;
        for (int a = 0; a < 3; ++a) b.lo[a] = n.lo[a][k], b.hi[a] = n.hi[a][k];
        return b.area();
    };
    Box rootb;
    for (int k = 0; k < 4; ++k)
        if (nodes[0].child[k] != RT_REF_NONE) {
            Box b;
            for (int a = 0; a < 3; ++a) b.lo[a] = nodes[0].lo[a][k], b.hi[a] = nodes[0].hi[a][k];
            rootb.add(b);
        }
    const double ra = rootb.area();
    if (!(ra > 0.0)) return 0.0;
    double sum = ra; // the root itself is always visited
    for (const Node4 &n : nodes)
        for (int k = 0; k < 4; ++k)
            if (n.child[k] != RT_REF_NONE && (n.child[k] >> RT_REF_KIND_SHIFT) == RT_KIND_NODE) sum += child_area(n, k);
    return 4.0 * sum / ra;
}

// What a closest-hit search through the tree costs, measured instead of modelled: the rays of a path
// tracer run from surface to surface, so segments between random points of two leaves' boxes (leaves
// drawn by surface area) are walked through the tree and what they touch is counted -- a node visit 1, a leaf by its kind's test
// relative to a visit (instruction counts of the device code).  The surface-area cost above assumes rays
// uniform through the root box; one far-out primitive (a fog around the world) inflates that box and the
// cost no longer tells two trees of the actual scene apart.  Deterministic.
inline double walk_cost(const std::vector<Node4> &nodes, uint32_t n_segments = 2048) {
    if (nodes.empty()) return 0.0;
    std::vector<Box> leaves;
    for (const Node4 &n : nodes)
        for (int k = 0; k < 4; ++k)
            if (n.child[k] != RT_REF_NONE && (n.child[k] >> RT_REF_KIND_SHIFT) != RT_KIND_NODE) {
                Box b;
                for (int a = 0; a < 3; ++a) b.lo[a] = n.lo[a][k], b.hi[a] = n.hi[a][k];
                leaves.push_back(b);
            }
    if (leaves.size() < 2) return 1.0;
    // a leaf is an end of a segment as often as its surface says: rays end on surfaces by their area
    std::vector<double> upto(leaves.size());
    double area_sum = 0.0;
    for (size_t i = 0; i < leaves.size(); ++i) upto[i] = (area_sum += std::max(leaves[i].area(), 1e-300));
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { // splitmix64 -> [0, 1)
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return double((z ^ (z >> 31)) >> 11) * (1.0 / 9007199254740992.0);
    };
    auto point_in = [&](const Box &b, double p[3]) {
        for (int a = 0; a < 3; ++a) p[a] = double(b.lo[a]) + rnd() * (double(b.hi[a]) - double(b.lo[a]));
    };
    const double leaf_weight[8] = {0.0, 0.5, 0.6, 1.6, 0.45, 0.0, 0.0, 0.0}; // by kind: quad, sphere, medium, box
    double total = 0.0;
    std::vector<uint32_t> stack;
    for (uint32_t i = 0; i < n_segments; ++i) {
        double o[3], e[3], inv[3];
        auto pick = [&]() { return std::min(size_t(std::lower_bound(upto.begin(), upto.end(), rnd() * area_sum) - upto.begin()), leaves.size() - 1); };
        point_in(leaves[pick()], o);
        point_in(leaves[pick()], e);
        for (int a = 0; a < 3; ++a) inv[a] = 1.0 / (e[a] - o[a]); // +-inf for a segment along an axis
        stack.assign(1, 0u);
        while (!stack.empty()) {
            const Node4 &n = nodes[stack.back()];
            stack.pop_back();
            total += 1.0;
            for (int k = 0; k < 4; ++k) {
                const uint32_t c = n.child[k];
                if (c == RT_REF_NONE) continue;
                double t0 = 0.0, t1 = 1.0;
                for (int a = 0; a < 3; ++a) {
                    double x0 = (double(n.lo[a][k]) - o[a]) * inv[a], x1 = (double(n.hi[a][k]) - o[a]) * inv[a];
                    if (x0 != x0 || x1 != x1) continue; // on a plane it runs along: no constraint
                    if (x0 > x1) std::swap(x0, x1);
                    t0 = std::max(t0, x0), t1 = std::min(t1, x1);
                }
                if (t0 > t1) continue;
                if ((c >> RT_REF_KIND_SHIFT) == RT_KIND_NODE) stack.push_back(c & RT_REF_INDEX_MASK);
                else total += leaf_weight[(c >> RT_REF_KIND_SHIFT) & 7u];
            }
        }
    }
    return total / n_segments;
}

// Builds the wide BVH of the (validated) scene.  builder: RT_BUILD_REFERENCE keeps the grouping
// of the descriptor's binary nodes, RT_BUILD_SAH regroups all primitives, RT_BUILD_AUTO builds both
// and keeps the one that is cheaper to walk (walk_cost).  boxes / box_faces: where the records of box
// leaves go (null: six quads stay six leaves); post: where primitives that enclose the scene go (null:
// they stay in the tree).  Returns "" or an error.
inline std::string build_wide_bvh(const rt_scene_desc *s, uint32_t builder, std::vector<Node4> &nodes, uint32_t &root,
                                  Stats &st, std::vector<WideBox> *boxes = nullptr, std::vector<uint32_t> *box_faces = nullptr,
                                  std::vector<uint32_t> *post = nullptr) {
    if (s->n_quads >= (1u << RT4_CLASS_SHIFT) || s->n_spheres >= (1u << RT4_CLASS_SHIFT) || s->n_media >= (1u << RT4_CLASS_SHIFT))
        return "more than 2^27 primitives of one kind";
    Builder b(s);
    b.find_boxes = boxes != nullptr; // a caller without room for the records gets quads only
    b.slots.reserve(size_t(s->n_quads + s->n_spheres + s->n_media) * 2 + 16);
    int32_t top = b.from_ref(s->root, 0);
    if (top < 0) return b.error;
    if (post) *post = b.take_enclosing(top);
    auto finish = [&](int32_t t, std::vector<Node4> &out, Stats &o) {
        o = Stats();
        root = b.collapse(t, out, o);
        o.stack_need = stack_need(out);
        o.cost = sa_cost(out);
        o.walk = walk_cost(out);
    };
    if (builder != RT_BUILD_SAH) finish(top, nodes, st);
    if (builder != RT_BUILD_REFERENCE && b.slots[top].node >= 0) {
        std::vector<int32_t> prims;
        b.collect_prims(top, prims);
        const int32_t sah_top = b.sah_tree(prims, 0, prims.size());
        std::vector<Node4> alt;
        Stats alt_st;
        finish(sah_top, alt, alt_st);
        if (builder == RT_BUILD_SAH || alt_st.walk < st.walk) nodes.swap(alt), st = alt_st;
    } else if (builder == RT_BUILD_SAH) {
        finish(top, nodes, st);
    }
    if (boxes) boxes->swap(b.boxes);
    if (box_faces) box_faces->swap(b.box_faces);
    return "";
}

// 128-byte nodes -> 64-byte nodes (same tree, same numbering).
inline void quantise_nodes(const std::vector<Node4> &nodes, std::vector<Node4Q> &out) {
    out.assign(nodes.size(), Node4Q{});
    for (size_t i = 0; i < nodes.size(); ++i) {
        const Node4 &n = nodes[i];
        Node4Q &q = out[i];
        for (int k = 0; k < 4; ++k) q.child[k] = n.child[k];
        for (int a = 0; a < 3; ++a) {
            double lo = std::numeric_limits<double>::infinity(), hi = -lo;
            for (int k = 0; k < 4; ++k)
                if (n.child[k] != RT_REF_NONE) lo = std::min(lo, double(n.lo[a][k])), hi = std::max(hi, double(n.hi[a][k]));
            if (!(lo <= hi)) lo = hi = 0.0; // a node without children (never built)
            // what fp32 rounding can move a plane by on the device, with room to spare
            const double margin = 8.0 * std::ldexp(std::max(std::fabs(lo), std::fabs(hi)), -23) + 1e-30;
            const float origin = std::nextafter(float(lo - margin), -std::numeric_limits<float>::infinity());
            const double extent = (hi + margin) - double(origin);
            int e = int(std::ceil(std::log2(std::max(extent, 1e-30) / 255.0)));
            while (std::ldexp(255.0, e) < extent) ++e; // log2 rounding
            e = std::min(std::max(e, -100), 100);
            const double step = std::ldexp(1.0, e);
            q.origin[a] = origin;
            q.exps |= uint32_t(e + 127) << (8 * a);
            for (int k = 0; k < 4; ++k) {
                uint32_t ql = 255u, qh = 0u; // an unused slot: inverted, and its ref says so too
                if (n.child[k] != RT_REF_NONE) {
                    const double l = std::floor((double(n.lo[a][k]) - margin - double(origin)) / step);
                    const double h = std::ceil((double(n.hi[a][k]) + margin - double(origin)) / step);
                    ql = uint32_t(std::min(std::max(l, 0.0), 255.0));
                    qh = uint32_t(std::min(std::max(h, 0.0), 255.0));
                }
                q.lo[a] |= ql << (8 * k);
                q.hi[a] |= qh << (8 * k);
            }
        }
    }
}

// Every quantised child box contains the fp32 one (with the rounding margin).  Returns "" or what is wrong.
inline std::string verify_quantised(const std::vector<Node4> &nodes, const std::vector<Node4Q> &q) {
    if (nodes.size() != q.size()) return "quantised node count differs";
    for (size_t i = 0; i < nodes.size(); ++i)
        for (int k = 0; k < 4; ++k) {
            if (q[i].child[k] != nodes[i].child[k]) return "quantised child refs differ";
            if (nodes[i].child[k] == RT_REF_NONE) continue;
            for (int a = 0; a < 3; ++a) {
                const double step = std::ldexp(1.0, int((q[i].exps >> (8 * a)) & 0xFFu) - 127);
                const double lo = double(q[i].origin[a]) + step * double((q[i].lo[a] >> (8 * k)) & 0xFFu);
                const double hi = double(q[i].origin[a]) + step * double((q[i].hi[a] >> (8 * k)) & 0xFFu);
                const double ulp = std::ldexp(std::max(std::fabs(double(nodes[i].lo[a][k])), std::fabs(double(nodes[i].hi[a][k]))), -23);
                if (!(lo <= double(nodes[i].lo[a][k]) - 4.0 * ulp && hi >= double(nodes[i].hi[a][k]) + 4.0 * ulp))
                    return "quantised box does not contain the fp32 box";
            }
        }
    return "";
}

// Structural check of a built tree against its scene: every primitive the descriptor's world
// reaches is a leaf exactly as often as the descriptor reaches it, every child box contains what
// is below it, indices point forward.  Returns "" or what is wrong.
inline std::string verify_wide_bvh(const rt_scene_desc *s, const std::vector<Node4> &nodes, const std::vector<WideBox> &boxes = {},
                                   const std::vector<uint32_t> &box_faces = {}, const std::vector<uint32_t> &post = {}) {
    if (nodes.empty()) return "no nodes";
    if (box_faces.size() != 6 * boxes.size()) return "box records and their face lists differ in number";
    // leaves the descriptor reaches (with multiplicity)
    std::vector<uint32_t> want;
    {
        std::vector<uint32_t> todo{s->root};
        while (!todo.empty()) {
            uint32_t r = todo.back();
            todo.pop_back();
            if ((r >> RT_REF_KIND_SHIFT) != RT_KIND_NODE) {
                want.push_back(r);
                continue;
            }
            const rt_node &n = s->nodes[r & RT_REF_INDEX_MASK];
            if (n.right == RT_NODE_LEAF)
                for (uint32_t k = 0; k < n.count; ++k) todo.push_back(s->refs[n.left + k]);
            else
                todo.push_back(n.left), todo.push_back(n.right);
        }
    }
    Builder b(s);
    b.boxes = boxes, b.box_faces = box_faces;
    std::vector<uint32_t> got;
    std::vector<Box> below(nodes.size());
    for (size_t i = nodes.size(); i-- > 0;) {
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = nodes[i].child[k];
            if (c == RT_REF_NONE) continue;
            Box stored;
            for (int a = 0; a < 3; ++a) stored.lo[a] = nodes[i].lo[a][k], stored.hi[a] = nodes[i].hi[a][k];
            Box content;
            if ((c >> RT_REF_KIND_SHIFT) == RT_KIND_NODE) {
                const uint32_t j = c & RT_REF_INDEX_MASK;
                if (j <= i || j >= nodes.size()) return "child index does not point forward";
                content = below[j];
            } else {
                const uint32_t leaf = c & ~RT4_CLASS_MASK;
                if ((leaf >> RT_REF_KIND_SHIFT) == RT_KIND_BOX) {
                    const uint32_t bi = leaf & RT_REF_INDEX_MASK;
                    if (bi >= boxes.size()) return "box leaf outside the box records";
                    // the record's planes and face refs are those of its six quads
                    const WideBox &wb = boxes[bi];
                    const uint32_t refs[6] = {wb.ref_lo_x, wb.ref_lo_y, wb.ref_lo_z, wb.ref_hi_x, wb.ref_hi_y, wb.ref_hi_z};
                    for (int f = 0; f < 6; ++f) {
                        const int axis = f % 3;
                        if ((refs[f] >> RT_REF_KIND_SHIFT) != RT_KIND_QUAD || (refs[f] & RT_REF_INDEX_MASK) >= s->n_quads) return "box face is not a quad";
                        if (std::find(&box_faces[6 * size_t(bi)], &box_faces[6 * size_t(bi)] + 6, refs[f]) == &box_faces[6 * size_t(bi)] + 6) return "box face not among the box's quads";
                        const rt_quad &q = s->quads[refs[f] & RT_REF_INDEX_MASK];
                        if (std::fabs(q.n[axis]) != 1.0f || q.q[axis] != (f < 3 ? wb.lo[axis] : wb.hi[axis])) return "box plane differs from its face's quad";
                    }
                    for (int f = 0; f < 6; ++f) got.push_back(box_faces[6 * size_t(bi) + f]);
                } else {
                    got.push_back(leaf);
                }
                if (c != b.with_class(leaf)) return "leaf does not carry the class of its material";
                content = b.prim_box(leaf);
            }
            for (int a = 0; a < 3; ++a)
                if (!(stored.lo[a] <= content.lo[a] && stored.hi[a] >= content.hi[a])) return "child box does not contain its content";
            below[i].add(stored);
        }
    }
    for (uint32_t r : post) { // tested after the walk instead of inside it
        if (r != b.with_class(r & ~RT4_CLASS_MASK)) return "a primitive tested after the walk does not carry the class of its material";
        got.push_back(r & ~RT4_CLASS_MASK);
    }
    std::sort(want.begin(), want.end());
    std::sort(got.begin(), got.end());
    if (want != got) return "leaves differ from the primitives the descriptor reaches";
    return "";
}

} // namespace accel
} // namespace rt
