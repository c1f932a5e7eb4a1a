"""Shared helpers for the parity tests: build one scene description in both the product's
host layer and the oracle, make fixed ray batches, compare images."""
import os

import numpy as np

import raytracing_amd as ra
from oracle import config_oracle
from oracle import oracle_py as orc


def build_both(scene_fn, **kw):
    """(host SceneBuilder, oracle SceneBuilder, CameraSettings) from one scene function."""
    host = ra.SceneBuilder()
    cam = scene_fn(host, **kw)
    osb = orc.scene()
    kw2 = dict(kw)
    if scene_fn in (ra.scenes.cornell_main, ra.scenes.cornell_book3):
        kw2["toml_objects"] = config_oracle.toml_objects
    cam2 = scene_fn(osb, **kw2)
    assert cam == cam2
    assert host.prim_count() == osb.prim_count()
    return host, osb, cam


def ray_batch(osb, cam, n, seed, bbox):
    """Fixed batch of fp32 rays: camera rays, bounces off their hit points (random, light-ward
    and grazing) and rays born inside the scene.  Returns (o, d, time, origin_prim): bounce rays
    carry the id of the primitive they start on, as the shade pass records it; -1 elsewhere."""
    rng = np.random.default_rng(seed)
    n_cam = n // 4
    W, H = cam.image_width, cam.image_height
    pix = rng.integers(0, W * H, n_cam).astype(np.uint32)
    strat = rng.integers(0, cam.effective_spp, n_cam).astype(np.uint32)
    o, d, tm = orc.camera_rays(osb, cam, pix, strat, seed=seed)
    o, d, tm = o.astype(np.float32), d.astype(np.float32), tm.astype(np.float32)
    hit, prim, t, u, v, t2 = orc.intersect_batch(osb, o, d, tm)
    ok = hit == 1
    p = (o.astype(np.float64) + t[:, None] * d.astype(np.float64))[ok].astype(np.float32)
    tmh, from_prim = tm[ok], prim[ok]
    k = len(p)
    dirs = rng.standard_normal((k, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    lo, hi = np.array(bbox[0]), np.array(bbox[1])
    targets = lo + rng.random((k, 3)) * (hi - lo)
    long_dirs = targets - p  # unnormalised, like directions towards a light sample
    graze = rng.standard_normal((k, 3))
    graze[np.arange(k), rng.integers(0, 3, k)] *= 1e-4
    inside_o = lo + rng.random((n_cam, 3)) * (hi - lo)
    inside_d = rng.standard_normal((n_cam, 3))
    none = -np.ones(n_cam, np.int32)
    O = np.concatenate([o, p, p, p, inside_o]).astype(np.float32)
    D = np.concatenate([d, dirs, long_dirs, graze, inside_d]).astype(np.float32)
    T = np.concatenate([tm, tmh, tmh, tmh, rng.random(n_cam)]).astype(np.float32)
    F = np.concatenate([none, from_prim, from_prim, from_prim, none]).astype(np.int32)
    good = np.all(np.isfinite(D), axis=1) & (np.abs(D).max(axis=1) > 0)
    return O[good], D[good], T[good], F[good]


def sphere_ids(host):
    d = host.flatten().contents
    return np.array([d.spheres[i].prim_id for i in range(d.n_spheres)], dtype=np.int64)


def check_intersection_parity(device, host, osb, cam, bbox, coord_scale, name="", n=65536, seed=3):
    """The bar of the fixed-ray-batch check on the scene uploaded to `device`: identical hit/miss
    and primitive id except where the oracle itself is unstable or two primitives tie, t within
    1e-5 relative (grazing hits: within 8 ulp of the coordinate scale from the oracle's surface),
    u and v within 2e-3."""
    o, d, tm, frm = ray_batch(osb, cam, n, seed=seed, bbox=bbox)
    assert len(o) > 0.9 * n
    hit, prim, t, u, v = device.intersect_batch(o, d, tm, origin_prim=frm)
    h2, p2, t2, u2, v2, second, nrm = orc.intersect_batch(osb, o, d, tm, want_normal=True)
    assert 0.2 < h2.mean() <= 1.0
    mism = np.nonzero((hit != h2) | (prim != p2))[0]
    if len(mism):
        unstable = unstable_rays(osb, o, d, tm, mism, coord_scale)
        # a mismatch is tolerated only where the oracle's own answer is unstable under
        # fp32-sized perturbations, or where two primitives are hit at the same t to fp32
        # resolution (coincident faces, shared edges)
        with np.errstate(invalid="ignore"):
            tie = np.abs(second[mism] - t2[mism]) <= 1e-5 * np.abs(t2[mism])
        both_hit = (hit[mism] == 1) & (h2[mism] == 1)
        near_t = both_hit & (np.abs(t[mism] - t2[mism]) <= 1e-5 * np.abs(t2[mism]))
        unexplained = ~(unstable | tie | near_t)
        assert not unexplained.any(), (name, mism[unexplained][:10], prim[mism][unexplained][:10], p2[mism][unexplained][:10],
                                       t[mism][unexplained][:10], t2[mism][unexplained][:10], o[mism][unexplained][:3],
                                       d[mism][unexplained][:3])
    # Coincident surfaces (neighbouring boxes share faces, a box stands on the floor) are met at
    # the same t: which of the two is reported depends on the visiting order, the reference's
    # being left child first.  Such rays must agree on t; everything else stays below 1e-3.
    n_tie = int((tie & near_t).sum()) if len(mism) else 0
    # A grazing ray off a plane whose coordinate fp32 cannot hold exactly (a box top at y = 94.9...):
    # the batch's fp32 origin sits a few 1e-6 off the plane, so the f64 oracle meets the plane it
    # starts on again beyond its 0.001 bound.  The device never re-tests the planar primitive a ray
    # leaves (in a render the oracle's own f64 origins are on the plane and it does not either);
    # it must then report what the oracle has as its second hit.
    n_rehit = 0
    if len(mism):
        with np.errstate(invalid="ignore"):
            gpu_is_second = np.where(np.isfinite(second[mism]), (hit[mism] == 1) & (np.abs(t[mism] - second[mism]) <= 1e-5 * np.abs(second[mism])),
                                     hit[mism] == 0)
        rehit = (frm[mism] >= 0) & (p2[mism] == frm[mism]) & gpu_is_second & ~(tie & near_t)
        n_rehit = int(rehit.sum())
    assert len(mism) - n_tie - n_rehit <= 1e-3 * len(o), (len(mism), n_tie, n_rehit)
    ok = np.nonzero((hit == 1) & (h2 == 1) & (prim == p2))[0]
    err = np.abs(t[ok] - t2[ok])
    rel = err / np.abs(t2[ok])
    # Bar: 1e-5 relative.  Where a ray meets a surface at grazing incidence, or starts within
    # fp32 resolution of the surface it hits, the along-ray error is the coordinate resolution
    # divided by |cos|: such hits are held to a distance from the oracle's hit surface of 8 ulp
    # of the coordinate scale, or must be as unstable in the oracle itself.
    dn = np.abs(np.einsum("ij,ij->i", nrm[ok], d[ok].astype(np.float64)))
    surf_dist = err * dn
    fine = (rel <= 1e-5) | (surf_dist <= 8 * 2.0 ** -23 * coord_scale)
    if not fine.all():
        left = np.nonzero(~fine)[0]
        wobble = unstable_t(osb, o, d, tm, ok[left], coord_scale)
        still = rel[left] > 4.0 * wobble
        assert not still.any(), (ok[left][still][:10], rel[left][still][:10], wobble[still][:10])
    general = frm[ok] != p2[ok]  # a ray re-entering the sphere it starts on is covered above
    # 1e-5 relative, or -- a short ray in a large scene: t is small against the coordinates its
    # hit point is made of -- a hit point within ONE fp32 ulp of the coordinate scale of the oracle's
    # surface; and on its own the 1e-5 bar holds for 99 % of the hits in every scene
    within = (rel <= 1e-5) | (surf_dist <= 2.0 ** -23 * coord_scale)
    assert within[general].mean() > 0.995, float(within[general].mean())
    assert (rel[general] <= 1e-5).mean() > 0.99, float((rel[general] <= 1e-5).mean())
    assert np.median(rel) < 2e-7
    uv_ok = ok[np.isfinite(u2[ok]) & (rel <= 1e-5)]
    du = np.abs(u[uv_ok] - u2[uv_ok])
    du = np.minimum(du, 1.0 - du)  # a sphere's u wraps at the atan2 seam (hittable.rs:155)
    on_sphere = np.isin(p2[uv_ok], sphere_ids(host))
    du = np.where(on_sphere, du * np.sin(np.pi * v2[uv_ok]), du)  # longitude is ill-conditioned at the poles
    assert du.max() < 2e-3 and np.abs(v[uv_ok] - v2[uv_ok]).max() < 2e-3
    assert np.median(du) < 1e-5
    return dict(rays=len(o), mismatches=int(len(mism)), same_t_ties=n_tie, origin_plane_rehits=n_rehit, median_rel_t=float(np.median(rel)), frac_t_within_1e5=float((rel[general] <= 1e-5).mean()))


def unstable_rays(osb, o, d, tm, idx, coord_scale, tmin=0.001, n_try=24, ulps=4.0, seed=0):
    """For rays idx: True where the ORACLE's own answer (hit, prim) flips when origin and
    direction move by a few fp32 ulps of the scene's coordinate scale -- edge grazes, ties and
    origins sitting on a surface, which no fp32 evaluation of the same inputs can be held to."""
    rng = np.random.default_rng(seed)
    o64, d64, t64 = o[idx].astype(np.float64), d[idx].astype(np.float64), tm[idx].astype(np.float64)
    h0, p0, *_ = orc.intersect_batch(osb, o64, d64, t64, tmin=tmin)
    flips = np.zeros(len(idx), bool)
    eps = ulps * 2.0 ** -23
    for _ in range(n_try):
        do = rng.standard_normal(o64.shape) * eps * coord_scale
        dd = d64 * (1.0 + rng.standard_normal(d64.shape) * eps)
        h, p, *_ = orc.intersect_batch(osb, o64 + do, dd, t64, tmin=tmin)
        flips |= (h != h0) | (p != p0)
    return flips


def unstable_t(osb, o, d, tm, idx, coord_scale, tmin=0.001, n_try=16, ulps=4.0, seed=0):
    """For rays idx: the largest change of the ORACLE's own t (relative) under the same
    perturbations; inf where its hit/primitive flips."""
    rng = np.random.default_rng(seed)
    o64, d64, t64 = o[idx].astype(np.float64), d[idx].astype(np.float64), tm[idx].astype(np.float64)
    h0, p0, t0, *_ = orc.intersect_batch(osb, o64, d64, t64, tmin=tmin)
    worst = np.zeros(len(idx))
    eps = ulps * 2.0 ** -23
    for _ in range(n_try):
        do = rng.standard_normal(o64.shape) * eps * coord_scale
        dd = d64 * (1.0 + rng.standard_normal(d64.shape) * eps)
        h, p, t, *_ = orc.intersect_batch(osb, o64 + do, dd, t64, tmin=tmin)
        with np.errstate(invalid="ignore", divide="ignore"):
            ch = np.where((h == h0) & (p == p0), np.abs(t - t0) / np.abs(t0), np.inf)
        worst = np.maximum(worst, np.nan_to_num(ch, nan=np.inf))
    return worst


def compare_same_stream(gpu_img, cpu_img):
    """Pixelwise comparison of two renders that consumed the same uniforms."""
    diff = np.abs(gpu_img.astype(np.float64) - cpu_img)
    rel = diff / (np.abs(cpu_img) + 1e-2)
    log = os.environ.get("RT_PARITY_LOG")  # measured margins of every same-stream comparison, one line each
    if log:
        with open(log, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}  {gpu_img.shape[1]}x{gpu_img.shape[0]}  "
                    f"median_rel={np.median(rel):.3e}  frac_1e3={(rel > 1e-3).mean():.4f}  frac_5e2={(rel > 5e-2).mean():.4f}\n")
    return dict(max_abs=float(diff.max()), median_rel=float(np.median(rel)), frac_1e3=float((rel > 1e-3).mean()),
                frac_5e2=float((rel > 5e-2).mean()), mean_gpu=gpu_img.mean(axis=(0, 1)), mean_cpu=cpu_img.mean(axis=(0, 1)))
