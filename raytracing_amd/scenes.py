"""The five workloads, written against SceneBuilder so that any library exporting the same
scene-construction calls can be driven from one description.

C1  cornell_main     main.rs:15-71 + cornell_box.toml: the only scene the reference ships
C2  book1_final      random spheres, defocus blur ("In One Weekend" final scene)
C3  cornell_book3    C1 geometry with a glass sphere for the cube, light + sphere in the light list
C4  book2_final      box field under a BVH, instanced sphere cloud, media, image + noise textures, motion blur
C5  stress           N random spheres/quads under one BVH
Scene randomness comes from numpy's Philox bit generator with a fixed key per scene.
"""
from pathlib import Path

import numpy as np

from .api import CameraSettings, SceneBuilder

_SCENES_DIR = Path(__file__).resolve().parent.parent / "scenes"


def cornell_toml_text():
    return (_SCENES_DIR / "cornell_box.toml").read_text()


def _cornell_camera(width, spp):
    cam = CameraSettings().with_aspect_ratio(width, 1.0)
    cam.max_depth = 50
    cam.samples_per_px = spp
    cam.background = (0.0, 0.0, 0.0)
    cam.camera_center = (278.0, 278.0, -800.0)
    cam.camera_target = (278.0, 278.0, 0.0)
    cam.vfov = 40.0
    cam.defocus_angle = 0.0
    return cam


def _cornell_walls(sb, toml_objects):
    """Returns (add, finish): add(obj) appends like world.add, finish() returns the world id."""
    text = cornell_toml_text()
    if toml_objects is None:
        world = sb.load_toml(text)  # ConfigModel::from_str + as_world
        return (lambda o: sb.list_add(world, o)), (lambda: world)
    objs = list(toml_objects(sb, text))
    return objs.append, (lambda: sb.list(objs))


def cornell_main(sb: SceneBuilder, width=600, spp=20, toml_objects=None):
    """main.rs:15-71.  toml_objects(sb, text) -> object ids, for builders without load_toml."""
    add, finish = _cornell_walls(sb, toml_objects)
    white = sb.lambertian((1.0, 1.0, 1.0))  # main.rs:32 (not the TOML's 0xBABABA)
    light = sb.light((20.0, 20.0, 20.0))    # main.rs:33-35
    lightbox = sb.quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), light)
    add(lightbox)
    box1 = sb.box3((0, 0, 0), (165.0, 330.0, 165.0), white)
    box1 = sb.translate(sb.rotate_y(box1, 15.0), (265.0, 0.0, 295.0))
    add(box1)
    box2 = sb.box3((0, 0, 0), (165.0, 165.0, 165.0), white)
    box2 = sb.translate(sb.rotate_y(box2, -18.0), (130.0, 0.0, 65.0))
    add(box2)
    sb.set_world(finish())
    sb.set_lights(lightbox)
    return _cornell_camera(width, spp)


def cornell_book3(sb: SceneBuilder, width=1200, spp=1000, toml_objects=None):
    """Cornell box of "The Rest of Your Life": glass sphere instead of the cube, and both the
    light and the sphere in the light list."""
    add, finish = _cornell_walls(sb, toml_objects)
    white = sb.lambertian((1.0, 1.0, 1.0))
    light = sb.light((20.0, 20.0, 20.0))
    glass = sb.dielectric(1.5)
    lightbox = sb.quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0), light)
    add(lightbox)
    box1 = sb.box3((0, 0, 0), (165.0, 330.0, 165.0), white)
    add(sb.translate(sb.rotate_y(box1, 15.0), (265.0, 0.0, 295.0)))
    ball = sb.sphere((190.0, 90.0, 190.0), 90.0, glass)
    add(ball)
    sb.set_world(finish())
    sb.set_lights(sb.list([lightbox, ball]))
    return _cornell_camera(width, spp)


def book1_final(sb: SceneBuilder, width=1200, spp=500, grid=11):
    rng = np.random.Generator(np.random.Philox(key=0xC0FFEE))
    objs = [sb.sphere((0.0, -1000.0, 0.0), 1000.0, sb.lambertian((0.5, 0.5, 0.5)))]
    for a in range(-grid, grid):
        for b in range(-grid, grid):
            choose = rng.random()
            center = (a + 0.9 * rng.random(), 0.2, b + 0.9 * rng.random())
            if np.linalg.norm(np.array(center) - np.array([4.0, 0.2, 0.0])) <= 0.9:
                continue
            if choose < 0.8:
                albedo = tuple(rng.random(3) * rng.random(3))
                mat = sb.lambertian(albedo)
            elif choose < 0.95:
                albedo = tuple(0.5 + 0.5 * rng.random(3))
                mat = sb.metal(albedo, 0.5 * rng.random())
            else:
                mat = sb.dielectric(1.5)
            objs.append(sb.sphere(center, 0.2, mat))
    objs.append(sb.sphere((0.0, 1.0, 0.0), 1.0, sb.dielectric(1.5)))
    objs.append(sb.sphere((-4.0, 1.0, 0.0), 1.0, sb.lambertian((0.4, 0.2, 0.1))))
    objs.append(sb.sphere((4.0, 1.0, 0.0), 1.0, sb.metal((0.7, 0.6, 0.5), 0.0)))
    sb.set_world(sb.list([sb.bvh(objs)]))
    sb.set_lights(None)
    cam = CameraSettings().with_aspect_ratio(width, 16.0 / 9.0)
    cam.samples_per_px, cam.max_depth = spp, 50
    cam.vfov = 20.0
    cam.camera_center, cam.camera_target = (13.0, 2.0, 3.0), (0.0, 0.0, 0.0)
    cam.defocus_angle, cam.focal_length = 0.6, 10.0
    return cam


EARTH_PNG = _SCENES_DIR / "textures" / "earth.png"


def earth_texture():
    """The reference's earth image (assets/textures/earth.png, 2048x1024 RGB8; the file under
    scenes/textures is that file byte for byte), decoded by the library's own PNG reader -- the
    loader behind ImageTexture::load (texture.rs:95-122)."""
    from . import imageio
    return imageio.read_png(EARTH_PNG)


def synthetic_earth(width=2048, height=1024):
    """Deterministic image of any size in the earth texture's format (RGB8): low-frequency
    'continents' over an ocean gradient.  For miniatures of the Book-2 scene in tests; the
    workload itself uses earth_texture()."""
    rng = np.random.Generator(np.random.Philox(key=0xEA27))
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    u, v = xx / width * 2 * np.pi, yy / height * np.pi
    f = np.zeros((height, width))
    for k in range(1, 7):
        a, b, c = rng.standard_normal(3)
        f += (a * np.sin(k * u + 3 * b) * np.sin(k * v + c) + b * np.cos((k + 1) * u - a) * np.sin((k + 2) * v)) / k
    land = f > 0.15
    img = np.empty((height, width, 3), np.uint8)
    img[..., 0] = np.where(land, 60 + 60 * np.clip(f, 0, 1), 10)
    img[..., 1] = np.where(land, 110 + 50 * np.clip(f, 0, 1), 40 + 30 * np.sin(v))
    img[..., 2] = np.where(land, 40, 120 + 80 * np.sin(v))
    return img


def perlin_tables(seed=0xBE71):
    rng = np.random.Generator(np.random.Philox(key=seed))
    vec = rng.random((256, 3)) * 2.0 - 1.0
    vec /= np.linalg.norm(vec, axis=1, keepdims=True)
    perm = np.stack([rng.permutation(256) for _ in range(3)]).astype(np.int32)
    return vec, perm


def book2_final(sb: SceneBuilder, width=800, spp=10000, boxes_per_side=20, cloud=1000, earth=None):
    rng = np.random.Generator(np.random.Philox(key=0xB00C2))
    ground = sb.lambertian((0.48, 0.83, 0.53))
    boxes1 = []
    w = 100.0
    for i in range(boxes_per_side):
        for j in range(boxes_per_side):
            x0, z0 = -1000.0 + i * w, -1000.0 + j * w
            y1 = 1.0 + 100.0 * rng.random()
            boxes1.append(sb.box3((x0, 0.0, z0), (x0 + w, y1, z0 + w), ground))
    world = [sb.bvh(boxes1)]
    light = sb.light((7.0, 7.0, 7.0))
    lightquad = sb.quad((123.0, 554.0, 147.0), (300.0, 0.0, 0.0), (0.0, 0.0, 265.0), light)
    world.append(lightquad)
    world.append(sb.sphere((400.0, 400.0, 200.0), 50.0, sb.lambertian((0.7, 0.3, 0.1)), center_vec=(30.0, 0.0, 0.0)))
    world.append(sb.sphere((260.0, 150.0, 45.0), 50.0, sb.dielectric(1.5)))
    world.append(sb.sphere((0.0, 150.0, 145.0), 50.0, sb.metal((0.8, 0.8, 0.9), 1.0)))
    boundary = sb.sphere((360.0, 150.0, 145.0), 70.0, sb.dielectric(1.5))
    world.append(boundary)
    world.append(sb.medium(boundary, 0.2, (0.2, 0.4, 0.9)))
    boundary2 = sb.sphere((0.0, 0.0, 0.0), 5000.0, sb.dielectric(1.5))
    world.append(sb.medium(boundary2, 0.0001, (1.0, 1.0, 1.0)))
    earth_img = earth_texture() if earth is None else earth
    world.append(sb.sphere((400.0, 200.0, 400.0), 100.0, sb.lambertian(sb.tex_image(earth_img))))
    vec, perm = perlin_tables()
    world.append(sb.sphere((220.0, 280.0, 300.0), 80.0, sb.lambertian(sb.tex_noise(0.2, vec, perm))))
    white = sb.lambertian((0.73, 0.73, 0.73))
    boxes2 = [sb.sphere(tuple(165.0 * rng.random(3)), 10.0, white) for _ in range(cloud)]
    world.append(sb.translate(sb.rotate_y(sb.bvh(boxes2), 15.0), (-100.0, 270.0, 395.0)))
    sb.set_world(sb.list(world))
    sb.set_lights(lightquad)
    cam = CameraSettings().with_aspect_ratio(width, 1.0)
    cam.samples_per_px, cam.max_depth = spp, 40
    cam.background = (0.0, 0.0, 0.0)
    cam.vfov = 40.0
    cam.camera_center, cam.camera_target = (478.0, 278.0, -600.0), (278.0, 278.0, 0.0)
    return cam


def stress(sb: SceneBuilder, n_prims=1_000_000, width=3840, height=2160, spp=4096, extent=100.0):
    rng = np.random.Generator(np.random.Philox(key=0x57E55))
    mats = [sb.lambertian(tuple(rng.random(3))) for _ in range(32)]
    metals = [sb.metal(tuple(0.5 + 0.5 * rng.random(3)), 0.3 * rng.random()) for _ in range(8)]
    glass = sb.dielectric(1.5)
    centers = (rng.random((n_prims, 3)) * 2.0 - 1.0) * extent
    kind = rng.random(n_prims)
    mat_pick = rng.random(n_prims)
    size = rng.random(n_prims)
    dirs = rng.standard_normal((n_prims, 2, 3))
    # quad edges for all primitives at once (a million per-primitive numpy calls take a minute)
    edge = 0.1 + 0.9 * size
    u_all = dirs[:, 0] / np.sqrt((dirs[:, 0] ** 2).sum(axis=1))[:, None] * edge[:, None]
    v_all = np.cross(u_all, dirs[:, 1])
    v_all = v_all / np.sqrt((v_all ** 2).sum(axis=1))[:, None] * edge[:, None]
    radius = 0.05 + 0.45 * size
    lam = np.minimum((mat_pick / 0.7 * 32).astype(np.int64), 10 ** 9) % 32
    met = ((mat_pick - 0.7) / 0.2 * 8).astype(np.int64) % 8
    centers_l, u_l, v_l = centers.tolist(), u_all.tolist(), v_all.tolist()
    objs = []
    for i in range(n_prims):
        m = mat_pick[i]
        mat = mats[lam[i]] if m < 0.7 else (metals[met[i]] if m < 0.9 else glass)
        if kind[i] < 0.5:
            objs.append(sb.sphere(centers_l[i], radius[i], mat))
        else:
            objs.append(sb.quad(centers_l[i], u_l[i], v_l[i], mat))
    sb.set_world(sb.list([sb.bvh(objs)]))
    sb.set_lights(None)
    cam = CameraSettings()
    cam.image_width, cam.image_height = width, height
    cam.samples_per_px, cam.max_depth = spp, 50
    cam.vfov = 40.0
    cam.camera_center, cam.camera_target = (0.0, 0.3 * extent, 2.6 * extent), (0.0, 0.0, 0.0)
    return cam
