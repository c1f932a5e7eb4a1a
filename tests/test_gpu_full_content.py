"""GPU parity at FULL scene content: the Book-1 final scene with all of its ~488 spheres, the
Book-2 final scene with its 400 boxes, 1000-sphere cloud and the reference's earth image, and the
stress scene at one million primitives (2 M binary nodes, 64 MB) -- the sizes the workloads are
defined at, where the miniatures of test_gpu_parity.py leave deep traversal stacks, large node
indices and the full material tables unexercised.

Intersection: the fixed-ray-batch bar against the oracle (65 k rays are seconds of CPU behind the
oracle's BVH).  Rendering at full resolution is beyond the oracle, so those tests use properties
that do not depend on size: exact sample counts, a strata split that sums to the one-pass film,
finite values, next to no dropped samples, a traversal stack within the LDS budget, and path
length and block means that agree with an oracle render of the same scene at low resolution."""
import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import distributed as rdist
from oracle import oracle_py as orc
import helpers

pytestmark = pytest.mark.gpu

# name: (scene function, kwargs = the workload as defined, box rays are aimed into, coordinate scale,
#        strata rendered at full resolution)
FULL = {
    "c2": (ra.scenes.book1_final, dict(width=1200, spp=500), ((-11, 0, -11), (11, 2, 11)), 1000.0, 22),
    "c4": (ra.scenes.book2_final, dict(width=800, spp=10000), ((-100, 0, -100), (600, 560, 600)), 1000.0, 16),
    "c5": (ra.scenes.stress, dict(n_prims=1_000_000), ((-100, -100, -100), (100, 100, 100)), 260.0, 1),
}
_built = {}


def full_scene(name):
    if name not in _built:
        fn, kw, bbox, scale, strata = FULL[name]
        _built[name] = helpers.build_both(fn, **kw)
    return _built[name]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", list(FULL))
def test_intersection_parity_at_full_content(device, name):
    host, osb, cam = full_scene(name)
    _, _, bbox, coord_scale, _ = FULL[name]
    d = device.upload(host)
    info = device.accel_info()
    assert info["kind"] == "bvh4"
    if name == "c2":
        assert d.n_spheres > 450
    if name == "c4":
        assert d.n_quads == 400 * 6 + 1 and d.n_spheres == 1000 + 8 and d.n_media == 2 and d.n_texel_bytes == 2048 * 1024 * 3
        # a box is one leaf of the tree searched; the fog around the world is tested after the walk; the fogs' boundary
        # spheres are not in the world themselves
        assert info["box_leaves"] == 400 and info["post_tests"] == 1 and info["leaves"] == 400 + 1 + 1006 + 1
    if name == "c5":
        assert d.n_quads + d.n_spheres == 1_000_000 and d.n_nodes >= 999_999
    # the per-lane stack lives in LDS: at most 64 entries fit a block at all, 40 keep four blocks on a CU
    assert 0 < info["stack_entries"] <= 40, info
    r = helpers.check_intersection_parity(device, host, osb, cam, bbox, coord_scale, name=name, seed=13)
    print(name, info, r)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", list(FULL))
def test_full_resolution_render_properties(device, name):
    host, osb, cam = full_scene(name)
    fn, kw, _, _, n_strata = FULL[name]
    device.upload(host)
    W, H = cam.image_width, cam.image_height
    device.film_resize(W, H)
    rc = cam.build()
    st = device.render(rc, mode=ra.RT_MODE_BOOK3, seed=1, stratum_begin=0, stratum_end=n_strata)
    whole = device.film().astype(np.float64)
    assert st.regrouped == (name == "c5")  # a pool of paths per wave behind the large tree, lane-bound paths behind the small ones
    if name == "c5":  # both variants of the persistent kernel trace the same samples at this size too
        device.film_clear()
        other = device.render(rc, mode=ra.RT_MODE_BOOK3, seed=1, stratum_begin=0, stratum_end=n_strata, regroup=False)
        assert not other.regrouped and other.camera_rays == st.camera_rays and abs(other.segments - st.segments) <= 1e-3 * st.segments
        # The two kernels are compiled apart: where the compiler packs or fuses the shared arithmetic
        # differently, a t differs in its last bit, and this scene's chains of bounces off mirror and glass
        # spheres of radius 0.05 blow that up -- as they do between the device and the f64 oracle
        # (test_render_same_stream_pixelwise[c5]; against the oracle at 300 k primitives either kernel has 14 %
        # of its values beyond 5 %, and the same mean to four digits).  Measured here, one sample per pixel:
        # most values equal to the bit, 5.6 % beyond 5 %, the frame's mean equal to 1e-4.
        lane = device.film().astype(np.float64)
        r = helpers.compare_same_stream(lane, whole)
        print("c5 lane-bound against pool kernel", r, lane.mean(axis=(0, 1)) / whole.mean(axis=(0, 1)))
        assert r["median_rel"] < 1e-6 and r["frac_5e2"] < 0.10, r
        assert np.allclose(lane.mean(axis=(0, 1)), whole.mean(axis=(0, 1)), rtol=2e-3)
    assert st.camera_rays == W * H * n_strata
    assert np.isfinite(whole).all() and whole.min() >= 0.0 and whole.max() > 0.0
    assert st.dropped <= 1e-6 * st.segments + 2, (st.dropped, st.segments)
    # the split of a multi-GPU run: three ranks' strata into one film = the one-pass film
    if n_strata >= 3:
        device.film_clear()
        seg = cam_rays = 0
        for rank in range(3):
            b, e = rdist.partition_strata(n_strata, 3, rank)
            s = device.render(rc, mode=ra.RT_MODE_BOOK3, seed=1, stratum_begin=b, stratum_end=e)
            seg, cam_rays = seg + s.segments, cam_rays + s.camera_rays
        parts = device.film().astype(np.float64)
        assert (seg, cam_rays) == (st.segments, st.camera_rays)
        assert np.allclose(parts, whole, rtol=1e-4, atol=1e-4 * n_strata)
    # an oracle render of the same scene at 1/25 of the resolution: path length and the picture's
    # block structure do not depend on resolution
    sw, sh, g = SMALL[name]  # same aspect ratio; g x g blocks divide both images evenly
    assert sw * H == sh * W and W % g == 0 and H % g == 0 and sw % g == 0 and sh % g == 0
    small = ra.CameraSettings(**{**cam.__dict__, "image_width": sw, "image_height": sh, "samples_per_px": 64})
    film, _, cnt = orc.render(osb, small, mode=orc.MODE_BOOK3, seed=5, threads=8)
    per_ray = (cnt["segments"] - cnt["zero_weight_segments"]) / cnt["camera_rays"]
    assert st.segments / st.camera_rays == pytest.approx(per_ray, rel=0.008)  # measured 1.4e-3, 3.8e-3, 1.5e-3, (st.segments / st.camera_rays, per_ray)
    gpu_img = whole / n_strata
    cpu_img = film / small.effective_spp
    gb = gpu_img.reshape(g, H // g, g, W // g, 3).mean(axis=(1, 3))
    cb = cpu_img.reshape(g, sh // g, g, sw // g, 3).mean(axis=(1, 3))
    corr = float(np.corrcoef(gb.ravel(), cb.ravel())[0, 1])
    ratio = float(gb.mean() / cb.mean())
    print(name, "segments/camera ray", st.segments / st.camera_rays, per_ray, "block corr", corr, "mean ratio", ratio)
    lo, hi = BLOCK_BOUNDS[name]
    assert corr > lo and abs(ratio - 1.0) < hi, (corr, ratio)


# block-mean agreement with the low-resolution oracle render: (correlation above, |mean ratio - 1| below).
# The Book-2 scene is lit by a small lamp through long paths: its block means are the noisiest.
SMALL = {"c2": (80, 45, 5), "c4": (64, 64, 8), "c5": (128, 72, 8)}
BLOCK_BOUNDS = {"c2": (0.995, 0.02), "c4": (0.9, 0.15), "c5": (0.99, 0.03)}


def test_main_rs_scene_at_the_shipped_size_against_the_reference_image(device):
    """The reference's own render: main.rs:15-26 builds a 600 x 600 camera with 20 -> 16 samples per
    pixel and depth 50, main.rs:28-71 the Cornell scene, camera.rs:455-470 is the shipped estimator and
    PngWriter writes linear 8-bit values (export.rs:77-81) -- assets/new-cornell.png, whose 20 x 20
    block means are committed as tests/golden/new_cornell_blockmeans.npy.  The device renders exactly
    that (RT_MODE_REF_EXACT through rt_film_quantize, the output stage the reference-side binding
    would call) and is held to the file with the bounds the oracle is held to
    (test_oracle_kats.py::test_reference_image_pins_shipped_estimator); camera-ray count and the
    strata split of a multi-GPU run are exact at this size too."""
    from pathlib import Path
    want = np.load(Path(__file__).resolve().parent / "golden" / "new_cornell_blockmeans.npy").astype(np.float64)
    sb = ra.SceneBuilder()
    cam = ra.scenes.cornell_main(sb)  # width 600, 20 samples asked for
    assert (cam.image_width, cam.image_height, cam.samples_per_px, cam.effective_spp, cam.max_depth) == (600, 600, 20, 16, 50)
    device.upload(sb)
    info = device.accel_info()
    assert info["kind"] == "flat" and (info["flat_boxes"], info["flat_rects"], info["flat_quads"]) == (3, 1, 0)  # room + two boxes, the lamp
    rc = cam.build()
    device.film_resize(600, 600)
    st = device.render(rc, mode=ra.RT_MODE_REF_EXACT, seed=3)
    assert st.camera_rays == 600 * 600 * 16 and st.regrouped
    whole = device.film().astype(np.float64)
    rgb8 = device.film_rgb8(1.0 / cam.effective_spp, gamma=False)  # what PngWriter would be handed
    assert rgb8.shape == (600, 600, 3) and rgb8.dtype == np.uint8
    got = (rgb8.astype(np.float64) / 255.0).reshape(20, 30, 20, 30, 3).mean(axis=(1, 3))
    corr = float(np.corrcoef(got.ravel(), want.ravel())[0, 1])
    print("new-cornell.png against the device: corr", corr, "mean ratio", got.mean() / want.mean(), "max block diff", np.abs(got - want).max())
    # the oracle's bounds are 0.995 / 5 % / 0.08 at a third of the resolution; at the reference's own size
    # and sample count the device measures 0.99996 / 0.07 % / 0.006
    assert corr > 0.9998
    assert abs(got.mean() / want.mean() - 1.0) < 0.005
    assert np.abs(got - want).max() < 0.015
    # the split of an eight-GPU run: eight stratum ranges into one film = the one-pass film
    device.film_clear()
    seg = cam_rays = 0
    for rank in range(8):
        b, e = rdist.partition_strata(cam.effective_spp, 8, rank)
        assert e - b == 2
        s = device.render(rc, mode=ra.RT_MODE_REF_EXACT, seed=3, stratum_begin=b, stratum_end=e)
        seg, cam_rays = seg + s.segments, cam_rays + s.camera_rays
    assert (seg, cam_rays) == (st.segments, st.camera_rays)
    assert np.allclose(device.film().astype(np.float64), whole, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["earth", "glass", "cornell-3", "cornell-4"])
def test_device_render_against_the_reference_held_images(device, name):
    """End to end against files the reference itself holds: the device's render of the scenes behind
    assets/earth.png, glass.png, cornell-3.png and cornell-4.png (tests/diary_scenes.py), through its
    own 8-bit output stage, against the block means of those images.  Same bounds as the oracle's
    pins in tests/test_oracle_kats.py."""
    import diary_scenes
    from test_oracle_kats import DIARY_BOUNDS, GOLD
    fn, kw, (rows, cols) = diary_scenes.DIARY[name]
    want = np.load(GOLD / "diary_blockmeans.npz")[name].astype(np.float64)
    host = ra.SceneBuilder()
    cam, gamma = fn(host, **kw)
    device.upload(host)
    device.film_resize(cam.image_width, cam.image_height)
    device.render(cam.build(), mode=ra.RT_MODE_BOOK3, seed=1)
    img = device.film_rgb8(1.0 / cam.effective_spp, gamma=gamma).astype(np.float64) / 255.0
    got = diary_scenes.block_means(img, rows, cols)
    corr, lo, hi = DIARY_BOUNDS[name]
    c = float(np.corrcoef(got.ravel(), want.ravel())[0, 1])
    print(name, "corr", c, "mean ratio", got.mean() / want.mean())
    assert c > corr and lo < got.mean() / want.mean() < hi, (c, got.mean() / want.mean())
