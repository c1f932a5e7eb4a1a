// rt_flat.h -- what the upload makes of a world that is one list of primitives (main.rs:30-69 builds
// such a world): the records of the stack-free scan (scan_flat in rt_kernels.h).  Host code only, no
// device calls: the upload uses it, and so does rt_accel_selftest, which needs no GPU.
#pragma once
#include "../../include/rt_amd.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

// how much of a box's frame is the world's own: the class word of a box record
enum : uint32_t { RT_BOX_GENERAL = 0, RT_BOX_UPRIGHT = 1, RT_BOX_ALIGNED = 2 };

namespace flat {

// Parallelograms that are faces of one rectangular box -- what box3 builds (hittable.rs:398-455),
// under any rigid map, and what the walls of a room are: a frame, the two plane offsets per axis
// measured from the centre, and the faces' refs by (axis, side).  n = 6 faces, or 5 with one face
// absent (RT_REF_NONE in its place: a room open towards the camera).  Every face present must span
// the box in its two directions, so the absent face's plane follows from the others.  Record
// layout: scan_flat.
inline bool match_box(const rt_quad *const *q, const uint32_t *refs, int n, float rec[32], bool general_frame = false) {
    if (n != 5 && n != 6) return false;
    typedef double D3[3];
    auto dot3 = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto len3 = [&](const double *a) { return std::sqrt(dot3(a, a)); };
    double extent = 0, reach = 0;
    for (int i = 0; i < n; ++i) {
        if (q[i]->shape != RT_SHAPE_PARALLELOGRAM) return false;
        D3 u = {q[i]->u[0], q[i]->u[1], q[i]->u[2]}, v = {q[i]->v[0], q[i]->v[1], q[i]->v[2]};
        const double lu = len3(u), lv = len3(v);
        if (!(lu > 0 && lv > 0) || std::fabs(dot3(u, v)) > 1e-6 * lu * lv) return false; // a rectangle
        for (int a = 0; a < 3; ++a) reach = std::max(reach, std::fabs(double(q[i]->q[a])));
        extent = std::max(extent, std::max(lu, lv));
    }
    const double tol = 1e-5 * (extent + reach); // the records are fp32: corners carry ~1e-7 relative
    D3 e[3];
    {
        const D3 u0 = {q[0]->u[0], q[0]->u[1], q[0]->u[2]}, v0 = {q[0]->v[0], q[0]->v[1], q[0]->v[2]};
        const double lu = len3(u0), lv = len3(v0);
        for (int a = 0; a < 3; ++a) e[0][a] = q[0]->u[a] / lu, e[1][a] = q[0]->v[a] / lv;
        e[2][0] = e[0][1] * e[1][2] - e[0][2] * e[1][1];
        e[2][1] = e[0][2] * e[1][0] - e[0][0] * e[1][2];
        e[2][2] = e[0][0] * e[1][1] - e[0][1] * e[1][0];
    }
    // every face in the frame: the axis of its normal, its plane there, its extent on the two other axes
    int axis[6];
    double plane[6], lo_of[6][3], hi_of[6][3];
    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < n; ++i) {
        D3 nrm = {q[i]->n[0], q[i]->n[1], q[i]->n[2]};
        int k = 0;
        for (int a = 1; a < 3; ++a)
            if (std::fabs(dot3(nrm, e[a])) > std::fabs(dot3(nrm, e[k]))) k = a;
        if (std::fabs(std::fabs(dot3(nrm, e[k])) - 1.0) > 1e-6) return false;
        axis[i] = k;
        for (int a = 0; a < 3; ++a) lo_of[i][a] = 1e300, hi_of[i][a] = -1e300;
        for (int c = 0; c < 4; ++c) {
            D3 p;
            for (int a = 0; a < 3; ++a) p[a] = double(q[i]->q[a]) + ((c & 1) ? double(q[i]->u[a]) : 0.0) + ((c & 2) ? double(q[i]->v[a]) : 0.0);
            for (int a = 0; a < 3; ++a) {
                const double x = dot3(p, e[a]);
                lo_of[i][a] = std::min(lo_of[i][a], x), hi_of[i][a] = std::max(hi_of[i][a], x);
            }
        }
        if (hi_of[i][k] - lo_of[i][k] > tol) return false; // its edges run along the two other axes
        plane[i] = 0.5 * (lo_of[i][k] + hi_of[i][k]);
        for (int a = 0; a < 3; ++a) blo[a] = std::min(blo[a], lo_of[i][a]), bhi[a] = std::max(bhi[a], hi_of[i][a]);
    }
    bool seen[3][2] = {};
    int face_at[3][2] = {{-1, -1}, {-1, -1}, {-1, -1}};
    for (int i = 0; i < n; ++i) {
        const int k = axis[i];
        if (!(bhi[k] - blo[k] > 2 * tol)) return false;
        const int side = std::fabs(plane[i] - blo[k]) <= tol ? 0 : (std::fabs(plane[i] - bhi[k]) <= tol ? 1 : -1);
        if (side < 0 || seen[k][side]) return false;
        for (int a = 0; a < 3; ++a) // the face spans the box in its two directions
            if (a != k && (std::fabs(lo_of[i][a] - blo[a]) > 2 * tol || std::fabs(hi_of[i][a] - bhi[a]) > 2 * tol)) return false;
        seen[k][side] = true;
        face_at[k][side] = i;
    }
    // The frame in canonical form.  Every rigid map of the reference is a RotateY and a Translate, so
    // one axis of a box is the world's y and the two others lie in the xz plane ("upright"); the
    // walls of a room are along all three world axes ("aligned").  Exact tests on the fp32 frame: a
    // frame that only nearly qualifies stays general.  Reordering axes and reversing one (its lo
    // and hi faces change places) leave the box what it is.
    auto unit_along = [&](int k, int w) { // +-1 when fp32(e[k]) is exactly +-the world axis w, else 0
        for (int a = 0; a < 3; ++a) {
            const float v = float(e[k][a]);
            if (a == w ? std::fabs(v) != 1.0f : v != 0.0f) return 0;
        }
        return e[k][w] > 0 ? 1 : -1;
    };
    int order[3] = {0, 1, 2}, sign[3] = {1, 1, 1};
    uint32_t cls = RT_BOX_GENERAL;
    for (int ky = 0; ky < 3 && cls == RT_BOX_GENERAL && !general_frame; ++ky) {
        if (!unit_along(ky, 1)) continue;
        int k0 = (ky + 1) % 3, k2 = (ky + 2) % 3;
        if (float(e[k0][1]) != 0.0f || float(e[k2][1]) != 0.0f) continue;
        cls = RT_BOX_UPRIGHT;
        if (unit_along(k0, 2) && unit_along(k2, 0)) std::swap(k0, k2);
        order[0] = k0, order[1] = ky, order[2] = k2;
        sign[1] = unit_along(ky, 1);
        if (unit_along(k0, 0) && unit_along(k2, 2)) cls = RT_BOX_ALIGNED, sign[0] = unit_along(k0, 0), sign[2] = unit_along(k2, 2);
    }
    double ce[3][3], clo[3], chi[3];
    int cface[3][2];
    for (int j = 0; j < 3; ++j) {
        const int k = order[j];
        for (int a = 0; a < 3; ++a) ce[j][a] = sign[j] * e[k][a];
        clo[j] = sign[j] > 0 ? blo[k] : -bhi[k], chi[j] = sign[j] > 0 ? bhi[k] : -blo[k];
        cface[j][0] = face_at[k][sign[j] > 0 ? 0 : 1], cface[j][1] = face_at[k][sign[j] > 0 ? 1 : 0];
    }
    std::fill(rec, rec + 32, 0.0f);
    double c[3] = {0, 0, 0}; // the centre, in world coordinates
    for (int j = 0; j < 3; ++j)
        for (int a = 0; a < 3; ++a) c[a] += 0.5 * (clo[j] + chi[j]) * ce[j][a];
    const uint32_t none = RT_REF_NONE;
    for (int j = 0; j < 3; ++j) {
        for (int a = 0; a < 3; ++a) rec[3 * j + a] = float(ce[j][a]);
        rec[9 + j] = float(c[j]);
        std::memcpy(&rec[18 + j], cface[j][0] >= 0 ? &refs[cface[j][0]] : &none, 4);
        std::memcpy(&rec[21 + j], cface[j][1] >= 0 ? &refs[cface[j][1]] : &none, 4);
    }
    std::memcpy(&rec[24], &cls, 4);
    // Planes, (lo, hi) per axis.  Along a world axis: the world coordinate, for a face present the
    // one its own quad record holds (the scan then repeats the quad test's arithmetic).  Along any
    // other axis: offsets from the centre as fp32 stores it, along the frame as fp32 stores it, a
    // face's own plane from its own quad record.  An absent face's plane comes from the box.
    for (int j = 0; j < 3; ++j) {
        const bool world_axis = cls == RT_BOX_ALIGNED || (cls == RT_BOX_UPRIGHT && j == 1);
        for (int side = 0; side < 2; ++side) {
            double plane_at = 0;
            const int i = cface[j][side];
            if (world_axis) {
                plane_at = i >= 0 ? double(q[i]->q[j]) : (side ? chi[j] : clo[j]);
            } else if (i >= 0) {
                for (int a = 0; a < 3; ++a) plane_at += (double(q[i]->q[a]) - double(rec[9 + a])) * double(rec[3 * j + a]);
            } else {
                double cc = 0;
                for (int a = 0; a < 3; ++a) cc += double(rec[9 + a]) * double(rec[3 * j + a]);
                plane_at = (side ? chi[j] : clo[j]) - cc;
            }
            rec[12 + 2 * j + side] = float(plane_at);
        }
    }
    return true;
}

// Would recognising these quads as a box change which primitive a tie goes to?  HittableVec::hit
// gives a tie to the later primitive (t <= closest); the scan tests the remaining quads first and
// the boxes after them, in list order.  That is the list's own order unless a remaining quad that
// comes after the box in the list lies in the plane of one of its faces.
inline bool quad_in_a_face_plane(const rt_quad &other, const rt_quad *const *faces, int n) {
    for (int i = 0; i < n; ++i) {
        const rt_quad &f = *faces[i];
        const double nn = double(other.n[0]) * f.n[0] + double(other.n[1]) * f.n[1] + double(other.n[2]) * f.n[2];
        if (std::fabs(std::fabs(nn) - 1.0) > 1e-6) continue;
        double d = 0, scale = 1.0;
        for (int a = 0; a < 3; ++a) d += (double(other.q[a]) - f.q[a]) * f.n[a], scale = std::max(scale, std::fabs(double(f.q[a])));
        if (std::fabs(d) <= 1e-5 * scale) return true;
    }
    return false;
}

// A rectangle along the world's axes -- a lamp under a ceiling, a floor: the normal is exactly +-a
// world axis and each edge runs exactly along one of the two others (tests on the fp32 record).  The
// quad test (hittable.rs:341-371) then needs no dot products: t = (plane - o_n) * rcp(d_n), and
// alpha, beta are one coordinate of the hit point each, scaled -- the same arithmetic the packed quad
// test does with the zeros left out, bit for bit.  The inside rule 0 <= alpha, beta <= 1 does not tell
// the two edges apart, so the in-plane axes are stored in ascending order.  Rectangles in one plane
// have one normal axis, so sorting the records by that axis keeps the list's order where ties arise.
//   record, 8 floats: plane, Q_a, Q_b, 1/extent_a (signed), 1/extent_b (signed), ref, -, normal axis
inline bool match_rect(const rt_quad &q, uint32_t ref, float rec[8]) {
#ifdef RT_NO_RECTS // experiments: every such rectangle stays a quad
    return false;
#endif
    if (q.shape != RT_SHAPE_PARALLELOGRAM) return false;
    auto only = [](const float *v) { // the one axis v has a component along, or -1
        int k = -1;
        for (int a = 0; a < 3; ++a)
            if (v[a] != 0.0f) k = k < 0 ? a : 3;
        return k < 3 ? k : -1;
    };
    const int kn = only(q.n), ka = only(q.avec), kb = only(q.bvec);
    if (kn < 0 || ka < 0 || kb < 0 || kn == ka || kn == kb || ka == kb) return false;
    if (std::fabs(q.n[kn]) != 1.0f || !std::isfinite(q.avec[ka]) || !std::isfinite(q.bvec[kb])) return false;
    const bool swap = ka > kb;
    const int a = swap ? kb : ka, b = swap ? ka : kb;
    rec[0] = q.q[kn], rec[1] = q.q[a], rec[2] = q.q[b];
    rec[3] = swap ? q.bvec[kb] : q.avec[ka], rec[4] = swap ? q.avec[ka] : q.bvec[kb];
    std::memcpy(&rec[5], &ref, 4);
    rec[6] = 0.0f;
    const uint32_t axis = uint32_t(kn);
    std::memcpy(&rec[7], &axis, 4);
    return true;
}

// A sphere of a flat world that nothing else reaches into: every other primitive of the list
// keeps at least the radius away from its centre (a rectangle by its true distance, other quads
// by their plane's, spheres by centre distance; any medium or motion in the list says no).  A
// ray that leaves such a sphere inwards meets the same sphere next, whatever else the list
// holds.  Touching counts as away: the ball's interior is open.
inline bool sphere_is_alone(const rt_scene_desc *s, const std::vector<uint32_t> &list, uint32_t self) {
    const rt_sphere &me = s->spheres[self & RT_REF_INDEX_MASK];
    if (me.moving || me.cv[0] != 0.f || me.cv[1] != 0.f || me.cv[2] != 0.f) return false;
    const double c[3] = {me.c0[0], me.c0[1], me.c0[2]}, r = me.radius;
    if (!(r > 0)) return false;
    for (uint32_t ref : list) {
        if (ref == self) continue;
        const uint32_t kind = ref >> RT_REF_KIND_SHIFT, idx = ref & RT_REF_INDEX_MASK;
        double dist = 0;
        if (kind == RT_KIND_QUAD) {
            const rt_quad &q = s->quads[idx];
            double rel[3], uu = 0, vv = 0, uv = 0, ru = 0, rv = 0, rn = 0;
            for (int a = 0; a < 3; ++a) {
                rel[a] = c[a] - q.q[a];
                uu += double(q.u[a]) * q.u[a], vv += double(q.v[a]) * q.v[a], uv += double(q.u[a]) * q.v[a];
                ru += rel[a] * q.u[a], rv += rel[a] * q.v[a], rn += rel[a] * q.n[a];
            }
            dist = std::fabs(rn); // the plane's distance bounds any shape's from below
            if (q.shape == RT_SHAPE_PARALLELOGRAM && uu > 0 && vv > 0 && std::fabs(uv) <= 1e-9 * std::sqrt(uu * vv)) {
                const double a = std::min(1.0, std::max(0.0, ru / uu)), b = std::min(1.0, std::max(0.0, rv / vv));
                double d2 = 0; // a rectangle: clamping the coordinates gives its closest point
                for (int k = 0; k < 3; ++k) {
                    const double e = rel[k] - a * q.u[k] - b * q.v[k];
                    d2 += e * e;
                }
                dist = std::sqrt(d2);
            }
        } else if (kind == RT_KIND_SPHERE) {
            const rt_sphere &o = s->spheres[idx];
            if (o.moving || o.cv[0] != 0.f || o.cv[1] != 0.f || o.cv[2] != 0.f) return false;
            double d2 = 0;
            for (int a = 0; a < 3; ++a) d2 += (c[a] - o.c0[a]) * (c[a] - o.c0[a]);
            dist = std::sqrt(d2) - o.radius;
        } else {
            return false;
        }
        if (!(dist >= r)) return false;
    }
    return true;
}

// The scan's input for one scene.  flat = false: the world is not one short list of primitives and
// everything else is empty.
struct Plan {
    bool flat = false;
    std::vector<float> quads;   // 16 floats per quad that stays a quad, list order, padded for the scan's read-ahead
    std::vector<float> pairs;   // the same quads two per 32 floats when all are parallelograms
    std::vector<float> boxes;   // 32 floats per recognised box, sorted by class (match_box)
    std::vector<float> rects;   // 8 floats per rectangle along the world's axes (match_rect): by normal axis, then list order
    uint32_t n_rects_x = 0, n_rects_y = 0; // of rects: normal along x (they come first), along y (next); the rest along z
    std::vector<uint32_t> other; // refs of everything that is not a quad, list order
    std::vector<float> spheres; // when everything that is not a quad is a sphere: 16 floats per sphere (centre, radius,
                                // centre's velocity, material, ref, 7 unused), list order, and `other` is left empty
    std::vector<uint32_t> alone_spheres; // refs of spheres nothing else reaches into (sphere_is_alone)
    uint32_t n_quads = 0, n_boxes_aligned = 0, n_boxes_upright = 0;
};

// Returns "" or what is wrong.
inline std::string plan(const rt_scene_desc *s, Plan &out) {
    out = Plan();
    std::vector<float> &flat_quads = out.quads, &flat_boxes = out.boxes, &flat_pairs = out.pairs;
    std::vector<uint32_t> &flat_other = out.other;
    std::vector<uint32_t> flat_quad_refs;
    {
        uint32_t rk = s->root >> RT_REF_KIND_SHIFT, ri = s->root & RT_REF_INDEX_MASK;
        if (rk == RT_KIND_NODE && s->nodes[ri].right == RT_NODE_LEAF && s->nodes[ri].count <= 64) {
            bool only_prims = true;
            for (uint32_t k = 0; k < s->nodes[ri].count; ++k)
                if ((s->refs[s->nodes[ri].left + k] >> RT_REF_KIND_SHIFT) == RT_KIND_NODE) only_prims = false;
            if (only_prims) {
                out.flat = true;
                for (uint32_t k = 0; k < s->nodes[ri].count; ++k) {
                    const uint32_t r = s->refs[s->nodes[ri].left + k];
                    if ((r >> RT_REF_KIND_SHIFT) != RT_KIND_QUAD) {
                        flat_other.push_back(r);
                        continue;
                    }
                    const rt_quad &q = s->quads[r & RT_REF_INDEX_MASK];
                    float rec[16] = {q.q[0], q.q[1], q.q[2], 0.f, q.n[0], q.n[1], q.n[2], 0.f,
                                     q.avec[0], q.avec[1], q.avec[2], 0.f, q.bvec[0], q.bvec[1], q.bvec[2], 0.f};
                    std::memcpy(&rec[3], &r, 4);
                    std::memcpy(&rec[7], &q.mat, 4);
                    std::memcpy(&rec[11], &q.shape, 4);
                    flat_quads.insert(flat_quads.end(), rec, rec + 16);
                    flat_quad_refs.push_back(r);
                }
                // Boxes at the end of the quads (main.rs:52-69 lists walls, light, boxes) leave the
                // quad records for one slab test each.  From the end: the scan takes quads first, then
                // boxes, and a later quad wins a tie (a box standing on the floor).
                struct BoxFaces {
                    const rt_quad *faces[6];
                    uint32_t refs[6];
                    int n;
                };
                std::vector<BoxFaces> found; // in list order
                while (flat_quad_refs.size() >= 6) {
                    const size_t first = flat_quad_refs.size() - 6;
                    BoxFaces b;
                    b.n = 6;
                    for (int i = 0; i < 6; ++i) b.refs[i] = flat_quad_refs[first + i], b.faces[i] = &s->quads[b.refs[i] & RT_REF_INDEX_MASK];
                    float rec[32];
                    if (!match_box(b.faces, b.refs, 6, rec)) break;
                    found.insert(found.begin(), b);
                    flat_quad_refs.resize(first);
                    flat_quads.resize(first * 16);
                }
                // A room: the quads the list starts with, when they are five or six faces of one box (the
                // walls of cornell_box.toml; the side towards the camera is open).  It is tested with the
                // boxes, before them as in the list; that keeps the list's tie order as long as no quad that
                // stays a quad lies in the plane of a wall.
                for (int n_faces = 6; n_faces >= 5; --n_faces) {
                    if (flat_quad_refs.size() < size_t(n_faces)) continue;
                    BoxFaces b;
                    b.n = n_faces;
                    for (int i = 0; i < n_faces; ++i) b.refs[i] = flat_quad_refs[i], b.faces[i] = &s->quads[b.refs[i] & RT_REF_INDEX_MASK];
                    float rec[32];
                    if (!match_box(b.faces, b.refs, n_faces, rec)) continue;
                    bool tie_order_kept = true;
                    for (size_t j = n_faces; j < flat_quad_refs.size(); ++j)
                        if (quad_in_a_face_plane(s->quads[flat_quad_refs[j] & RT_REF_INDEX_MASK], b.faces, n_faces)) tie_order_kept = false;
                    if (!tie_order_kept) break;
                    found.insert(found.begin(), b);
                    flat_quad_refs.erase(flat_quad_refs.begin(), flat_quad_refs.begin() + n_faces);
                    flat_quads.erase(flat_quads.begin(), flat_quads.begin() + size_t(n_faces) * 16);
                    break;
                }
                // Rectangles along the world's axes leave the quad records too.  The scan tests them after
                // the quads that stay quads and before the boxes, in list order among themselves: the
                // list's own tie order, unless a quad that stays a quad comes later in the list and
                // lies in the rectangle's plane -- then the rectangle stays a quad as well.
                {
                    std::vector<char> take(flat_quad_refs.size(), 0);
                    for (size_t i = flat_quad_refs.size(); i-- > 0;) {
                        const rt_quad *me = &s->quads[flat_quad_refs[i] & RT_REF_INDEX_MASK];
                        float rec[8];
                        if (!match_rect(*me, flat_quad_refs[i], rec)) continue;
                        bool tie_order_kept = true;
                        for (size_t j = i + 1; j < flat_quad_refs.size(); ++j)
                            if (!take[j] && quad_in_a_face_plane(s->quads[flat_quad_refs[j] & RT_REF_INDEX_MASK], &me, 1)) tie_order_kept = false;
                        take[i] = tie_order_kept;
                    }
                    std::vector<float> kept_quads, by_axis[3]; // the scan runs one loop per normal axis
                    std::vector<uint32_t> kept_refs;
                    for (size_t i = 0; i < flat_quad_refs.size(); ++i) {
                        if (take[i]) {
                            float rec[8];
                            (void)match_rect(s->quads[flat_quad_refs[i] & RT_REF_INDEX_MASK], flat_quad_refs[i], rec);
                            uint32_t axis;
                            std::memcpy(&axis, &rec[7], 4);
                            by_axis[axis].insert(by_axis[axis].end(), rec, rec + 8);
                        } else {
                            kept_quads.insert(kept_quads.end(), flat_quads.begin() + 16 * i, flat_quads.begin() + 16 * (i + 1));
                            kept_refs.push_back(flat_quad_refs[i]);
                        }
                    }
                    for (int a = 0; a < 3; ++a) out.rects.insert(out.rects.end(), by_axis[a].begin(), by_axis[a].end());
                    out.n_rects_x = uint32_t(by_axis[0].size() / 8), out.n_rects_y = uint32_t(by_axis[1].size() / 8);
                    flat_quads.swap(kept_quads);
                    flat_quad_refs.swap(kept_refs);
                }
                // The scan runs the box classes one after the other (aligned, upright, general), and
                // between boxes too a later one wins a tie.  When the list's own order is another, every
                // box keeps a general frame and the list's order.
                for (int pass = 0; pass < 2; ++pass) {
                    flat_boxes.clear();
                    out.n_boxes_aligned = out.n_boxes_upright = 0;
                    bool in_class_order = true;
                    uint32_t last = RT_BOX_ALIGNED;
                    for (const BoxFaces &b : found) {
                        float rec[32];
                        if (!match_box(b.faces, b.refs, b.n, rec, pass == 1)) return "a recognised box no longer matches";
                        uint32_t cls;
                        std::memcpy(&cls, &rec[24], 4);
                        if (cls > last) in_class_order = false;
                        last = cls;
                        out.n_boxes_aligned += cls == RT_BOX_ALIGNED, out.n_boxes_upright += cls == RT_BOX_UPRIGHT;
                        flat_boxes.insert(flat_boxes.end(), rec, rec + 32);
                    }
                    if (in_class_order) break;
                }
                const std::vector<uint32_t> list(s->refs + s->nodes[ri].left, s->refs + s->nodes[ri].left + s->nodes[ri].count);
                for (uint32_t r : list)
                    if ((r >> RT_REF_KIND_SHIFT) == RT_KIND_SPHERE && sphere_is_alone(s, list, r)) out.alone_spheres.push_back(r);
            }
        }
    }
    // Spheres only among the rest (no medium): their records in list order, so that the scan reads a
    // sphere with one request instead of a ref, a decision on its kind and then the record.
    {
        bool all_spheres = !flat_other.empty();
        for (uint32_t r : flat_other) all_spheres = all_spheres && (r >> RT_REF_KIND_SHIFT) == RT_KIND_SPHERE;
        if (all_spheres) {
            for (uint32_t r : flat_other) {
                const rt_sphere &sp = s->spheres[r & RT_REF_INDEX_MASK];
                float rec[16] = {sp.c0[0], sp.c0[1], sp.c0[2], sp.radius, sp.cv[0], sp.cv[1], sp.cv[2]};
                std::memcpy(&rec[7], &sp.mat, 4);
                std::memcpy(&rec[8], &r, 4);
                out.spheres.insert(out.spheres.end(), rec, rec + 16);
            }
            flat_other.clear();
        }
    }
    // parallelograms only: the same records two per 32 floats, fields interleaved
    {
        const size_t nq = flat_quads.size() / 16;
        bool all_par = nq > 0;
        for (size_t i = 0; i < nq; ++i) {
            uint32_t shape;
            std::memcpy(&shape, &flat_quads[16 * i + 11], 4);
            all_par = all_par && shape == RT_SHAPE_PARALLELOGRAM;
        }
        if (all_par) {
            flat_pairs.assign((nq + 1) / 2 * 32, 0.0f); // an odd count leaves a zero normal: never hit
            for (size_t i = 0; i < nq; ++i) {
                const float *q = &flat_quads[16 * i];
                float *pr = &flat_pairs[(i / 2) * 32];
                const size_t j = i & 1;
                const int src[12] = {0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14}; // q, n, avec, bvec
                for (int f = 0; f < 12; ++f) pr[2 * f + j] = q[src[f]];
                pr[24 + j] = q[3]; // ref bits
                pr[26 + j] = q[7]; // material bits
            }
            const uint32_t none = RT_REF_NONE;
            if (nq & 1) std::memcpy(&flat_pairs[(nq / 2) * 32 + 25], &none, 4);
        }
    }
    out.n_quads = uint32_t(flat_quads.size() / 16);
    // the scan requests records in pairs, one pair ahead: pad with records that cannot hit
    flat_quads.resize((size_t(out.n_quads) + 3) / 2 * 2 * 16, 0.0f);
    return "";
}

} // namespace flat
