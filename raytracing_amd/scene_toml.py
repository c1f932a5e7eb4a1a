"""Scene files: a whole render (world, lights, camera) as TOML.

The format is the reference's cornell_box.toml (config.rs) extended as described in
raytracing_amd/host/config.cpp.  `TomlRecorder` has SceneBuilder's construction calls and writes
what it is told to build as such a file, so the scene functions of `scenes.py` double as the
generators of the data files under scenes/.  Loading goes through the library
(`load_scene`, rth_load_scene_toml).
"""
import ctypes as C
import math
from pathlib import Path

import numpy as np

from . import _ffi, imageio
from .api import CameraSettings, SHAPE_DISC, SHAPE_PARALLELOGRAM, SHAPE_TRIANGLE


def camera_from_struct(b) -> CameraSettings:
    cam = CameraSettings()
    cam.image_width, cam.image_height = b.image_width, b.image_height
    cam.vfov, cam.samples_per_px, cam.max_depth = b.vfov_deg, b.samples_per_px, b.max_depth
    cam.background = None if b.background_sky else tuple(b.background)
    cam.camera_center, cam.camera_target, cam.vup = tuple(b.center), tuple(b.target), tuple(b.vup)
    cam.defocus_angle, cam.focal_length = b.defocus_angle_deg, b.focal_length
    cam.antialiasing = "disc" if b.antialiasing == 1 else "square"
    return cam


def load_scene(sb, text, base_dir=""):
    """Builds the file's world and lights in SceneBuilder `sb` (rth_load_scene_toml) and
    returns its CameraSettings, or None when the file has no [camera] table."""
    cam, has = _ffi.CameraBuilderStruct(), C.c_int()
    sb.world = sb._call("load_scene_toml", C.c_char_p(text.encode()), C.c_char_p(str(base_dir).encode()), C.byref(cam), C.byref(has))
    return camera_from_struct(cam) if has.value else None


def load_scene_file(sb, path):
    path = Path(path)
    return load_scene(sb, path.read_text(), path.parent)


# ---------------------------------------------------------------------------- writing
def _num(x):
    x = float(x)
    if not math.isfinite(x):
        raise ValueError("scene files hold finite numbers only")
    r = repr(x)
    return r if ("." in r or "e" in r or "n" in r) else r + ".0"


def _vec(v):
    return "[" + ", ".join(_num(x) for x in v) + "]"


def _val(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    if isinstance(v, (float, np.floating)):
        return _num(v)
    if isinstance(v, str):
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, dict):
        return "{ " + ", ".join(f"{k} = {_val(x)}" for k, x in v.items()) + " }"
    if isinstance(v, (list, tuple, np.ndarray)):
        return "[" + ", ".join(_val(x) for x in v) + "]"
    raise TypeError(type(v))


class _Node:
    def __init__(self, kind, **fields):
        self.kind, self.fields = kind, fields
        self.light = False
        self.name = None
        self.emitted = False


class TomlRecorder:
    """SceneBuilder's construction interface; records instead of building."""

    def __init__(self, shared_images=False):
        # shared_images: an image equal to the earth texture is referenced at ../textures/earth.png
        # (files written under scenes/<name>/) instead of being written beside the scene file
        self.shared_images = shared_images
        self.textures, self.materials, self.objects = [], [], []
        self.images = {}  # texture index -> uint8 array
        self.world = self.lights = None
        self.light_list = False
        self._solid_cache = {}

    # textures
    def tex_solid(self, r, g, b):
        self.textures.append(dict(type="color", color=[float(r), float(g), float(b)]))
        return len(self.textures) - 1

    def tex_checker(self, scale, even, odd):
        self.textures.append(dict(type="checkerboard", scale=float(scale), textures=[self._tname(even), self._tname(odd)]))
        return len(self.textures) - 1

    def tex_image(self, rgb):
        i = len(self.textures)
        img = np.ascontiguousarray(rgb, dtype=np.uint8)
        if self.shared_images:
            from . import scenes
            if img.shape == (1024, 2048, 3) and np.array_equal(img, scenes.earth_texture()):
                # the earth image lives once under scenes/textures; scene files sit one level below scenes/
                self.textures.append(dict(type="image", path="../textures/earth.png"))
                return i
        self.images[i] = img
        self.textures.append(dict(type="image", path=f"texture_{i:04d}.png"))
        return i

    def tex_noise(self, scale, ranvec, perm):
        rv = np.asarray(ranvec, dtype=np.float64).reshape(256, 3)
        pm = np.asarray(perm, dtype=np.int64).reshape(3, 256)
        self.textures.append(dict(type="noise", scale=float(scale), ranvec=[[float(x) for x in v] for v in rv],
                                  perm=[[int(x) for x in row] for row in pm]))
        return len(self.textures) - 1

    @staticmethod
    def _tname(i):
        return f"t{int(i):04d}"

    def _tex_field(self, t):
        if isinstance(t, (tuple, list)):
            return dict(color=[float(x) for x in t])
        return dict(texture=self._tname(t))

    # materials
    def _mat(self, **fields):
        self.materials.append(fields)
        return len(self.materials) - 1

    def lambertian(self, t):
        return self._mat(type="Lambertian", **self._tex_field(t))

    def light(self, t):
        return self._mat(type="DiffuseLight", **self._tex_field(t))

    def metal(self, color, fuzz):
        return self._mat(type="Metal", albedo=[float(x) for x in color], fuzz=float(fuzz))

    def dielectric(self, ior):
        return self._mat(type="Dielectric", refractive_index=float(ior))

    def isotropic(self, t):
        return self._mat(type="Isotropic", **self._tex_field(t))

    def _mname(self, m):
        return m if isinstance(m, str) else f"m{int(m):04d}"

    # objects
    def _obj(self, kind, **fields):
        self.objects.append(_Node(kind, **fields))
        return len(self.objects) - 1

    def sphere(self, center, radius, mat, center_vec=(0.0, 0.0, 0.0)):
        f = dict(center=list(map(float, center)), radius=float(radius), material=self._mname(mat))
        if any(float(x) != 0.0 for x in center_vec):
            f["center_vec"] = list(map(float, center_vec))
        return self._obj("Sphere", **f)

    def quad(self, corner, u, v, mat, shape=SHAPE_PARALLELOGRAM):
        corner, u, v = (np.asarray(x, dtype=np.float64) for x in (corner, u, v))
        m = self._mname(mat)
        if shape == SHAPE_PARALLELOGRAM:
            return self._obj("Parallelogram", corner=list(corner), vectors=[list(u), list(v)], material=m)
        if shape == SHAPE_TRIANGLE:
            return self._obj("Triangle", points=[list(corner), list(corner + u), list(corner + v)], material=m)
        raise ValueError("discs are written from their centre: use disc()")

    def box3(self, a, b, mat):
        return self._obj("Box", min=list(map(float, a)), max=list(map(float, b)), material=self._mname(mat))

    def list(self, objs):
        return self._obj("Group", children=[int(o) for o in objs], bvh=False)

    def bvh(self, objs):
        return self._obj("Group", children=[int(o) for o in objs], bvh=True)

    def translate(self, obj, offset):
        return self._obj("translate", child=int(obj), offset=list(map(float, offset)))

    def rotate_y(self, obj, degrees):
        return self._obj("rotate_y", child=int(obj), degrees=float(degrees))

    def medium(self, boundary, density, t):
        return self._obj("ConstantMedium", boundary=int(boundary), density=float(density), **self._tex_field(t))

    def set_world(self, obj):
        self.world = int(obj)

    def set_lights(self, obj):
        self.lights = None if obj is None else int(obj)

    def list_add(self, lst, obj):
        self.objects[lst].fields["children"].append(int(obj))

    def load_toml(self, text):
        """The reference's own format (textures / materials / objects): merged in, names kept."""
        import tomli  # only this generator path needs a TOML reader in Python
        d = tomli.loads(text)
        if d.get("textures"):
            raise ValueError("TomlRecorder.load_toml: named textures of the source file are not carried over")
        self.named_materials = dict(d.get("materials", {}))
        ids = []
        for o in d.get("objects", []):
            o = dict(o)
            ids.append(self._obj(o.pop("type"), **o))
        return self.list(ids)

    # ------------------------------------------------------------------ output
    def _emit(self, i, uses):
        """Inline-table form of object i."""
        n = self.objects[i]
        if n.name is not None and n.emitted:
            return n.name
        if n.kind == "translate":
            inner = self._emit(n.fields["child"], uses)
            if isinstance(inner, str) or "translate" in inner or "name" in inner or "light" in inner:
                inner = dict(type="Group", children=[inner])
            inner["translate"] = n.fields["offset"]
            out = inner
        elif n.kind == "rotate_y":
            inner = self._emit(n.fields["child"], uses)
            if isinstance(inner, str) or "translate" in inner or "rotate_y" in inner or "name" in inner or "light" in inner:
                inner = dict(type="Group", children=[inner])
            inner["rotate_y"] = n.fields["degrees"]
            out = inner
        elif n.kind == "Group":
            out = dict(type="Group", children=[self._emit(c, uses) for c in n.fields["children"]])
            if n.fields.get("bvh"):
                out["bvh"] = True
        elif n.kind == "ConstantMedium":
            f = dict(n.fields)
            out = dict(type="ConstantMedium", boundary=self._emit(f.pop("boundary"), uses), **f)
        else:
            out = dict(type=n.kind, **n.fields)
        if n.light:
            out["light"] = True
        if uses.get(i, 0) > 1:
            n.name = f"o{i:05d}"
            out["name"] = n.name
        n.emitted = True
        return out

    def dumps(self, cam: CameraSettings = None):
        if self.world is None:
            raise ValueError("no world set")
        for n in self.objects:
            n.light, n.name, n.emitted = False, None, False
        lights = []
        if self.lights is not None:
            ln = self.objects[self.lights]
            lights = list(ln.fields["children"]) if ln.kind == "Group" else [self.lights]
            self.light_list = ln.kind == "Group"
        for l in lights:
            self.objects[l].light = True
        # how often each object is referenced from the world (shared objects get a name)
        uses = {}

        def count(i):
            uses[i] = uses.get(i, 0) + 1
            if uses[i] > 1:
                return
            n = self.objects[i]
            for c in n.fields.get("children", []):
                count(c)
            for k in ("child", "boundary"):
                if k in n.fields:
                    count(n.fields[k])

        top = self.objects[self.world]
        roots = list(top.fields["children"]) if top.kind == "Group" and not top.fields.get("bvh") else [self.world]
        for r in roots:
            count(r)
        out = ["# written by raytracing_amd.scene_toml.TomlRecorder\n"]
        if cam is not None:
            out.append("[camera]")
            out.append(f"image_width = {cam.image_width}\nimage_height = {cam.image_height}")
            out.append(f"samples_per_px = {cam.samples_per_px}\nmax_depth = {cam.max_depth}\nvfov = {_num(cam.vfov)}")
            out.append(f"camera_center = {_vec(cam.camera_center)}\ncamera_target = {_vec(cam.camera_target)}\nvup = {_vec(cam.vup)}")
            out.append(f"defocus_angle = {_num(cam.defocus_angle)}\nfocal_length = {_num(cam.focal_length)}")
            if cam.antialiasing != "square":
                out.append(f'antialiasing = "{cam.antialiasing}"')
            out.append('background = "sky"\n' if cam.background is None else f"background = {_vec(cam.background)}\n")
        if self.light_list:
            out.insert(1, "light_list = true\n")
        for i, t in enumerate(self.textures):
            out.append(f"[textures.{self._tname(i)}]")
            out += [f"{k} = {_val(v)}" for k, v in t.items()]
            out.append("")
        if not self.textures:
            out.append("textures = {}\n")
        for name, m in getattr(self, "named_materials", {}).items():
            out.append(f"[materials.{name}]")
            out += [f"{k} = {_val(v)}" for k, v in m.items()]
            out.append("")
        for i, m in enumerate(self.materials):
            out.append(f"[materials.{self._mname(i)}]")
            out += [f"{k} = {_val(v)}" for k, v in m.items()]
            out.append("")
        for r in roots:
            e = self._emit(r, uses)
            if isinstance(e, str):
                e = dict(type="Group", children=[e])
            out.append("[[objects]]")
            for k, v in e.items():
                if k == "children":  # one child per line keeps big groups readable and diffable
                    out.append("children = [\n" + ",\n".join("  " + _val(c) for c in v) + "\n]")
                else:
                    out.append(f"{k} = {_val(v)}")
            out.append("")
        return "\n".join(out)

    def save(self, path, cam: CameraSettings = None):
        path = Path(path)
        path.write_text(self.dumps(cam))
        for i, img in self.images.items():
            imageio.write_png(path.parent / self.textures[i]["path"], img)
