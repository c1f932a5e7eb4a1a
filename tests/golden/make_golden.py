"""Regenerates the fixtures in this directory.  Run from the repository root:
    python tests/golden/make_golden.py
* new_cornell_blockmeans.npy comes from the reference's own output image
  assets/new-cornell.png (600x600 linear PNG written by its PngWriter): 20x20 block means of
  the 8-bit values / 255.  Needs /root/reference; skipped when it is absent.
* diary_blockmeans.npz: block means (8-bit values / 255) of four more images the reference holds --
  assets/earth.png, glass.png (400x225, 9x16 blocks), cornell-3.png, cornell-4.png (600x600, 20x20
  blocks) -- for the pins in tests/test_oracle_kats.py.  Needs /root/reference; skipped when absent.
* the rest are outputs of the CPU oracle (oracle/liboracle.so) on fixed seeds.
* c?_oracle_64x64_converged.npz: the 64x64 frame at 16384 spp (128^2 strata), plus the oracle's
  own RMSE against it at 256 spp for three seeds of the device's generator (so a device render
  with the same seed can be held to "no worse than 1.1x the CPU render at equal spp").
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
HERE = Path(__file__).resolve().parent

import raytracing_amd as ra  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402
import helpers  # noqa: E402


def main():
    ref_png = Path("/root/reference/assets/new-cornell.png")
    if ref_png.exists():
        from PIL import Image
        a = np.asarray(Image.open(ref_png)).astype(np.float64) / 255.0
        bm = a.reshape(20, 30, 20, 30, 3).mean(axis=(1, 3)).astype(np.float32)
        np.save(HERE / "new_cornell_blockmeans.npy", bm)
    assets = Path("/root/reference/assets")
    if assets.exists():
        from PIL import Image
        import diary_scenes
        out = {}
        for name, (_, _, (rows, cols)) in diary_scenes.DIARY.items():
            a = np.asarray(Image.open(assets / f"{name}.png").convert("RGB")).astype(np.float64) / 255.0
            out[name] = diary_scenes.block_means(a, rows, cols).astype(np.float32)
        np.savez_compressed(HERE / "diary_blockmeans.npz", **out)
    if "--images-only" in sys.argv:
        return
    for name, fn, mode in (("c1", ra.scenes.cornell_main, orc.MODE_REF_EXACT), ("c3", ra.scenes.cornell_book3, orc.MODE_BOOK3)):
        host, osb, cam = helpers.build_both(fn, width=64, spp=16)
        film, _, cnt = orc.render(osb, cam, mode=mode, seed=1)
        np.save(HERE / f"{name}_oracle_64x64_spp16.npy", (film / cam.effective_spp).astype(np.float32))
        o, d, tm, frm = helpers.ray_batch(osb, cam, 4096, seed=7, bbox=((0, 0, 0), (555, 555, 555)))
        hit, prim, t, u, v, t2 = orc.intersect_batch(osb, o, d, tm)
        conv_cam = ra.CameraSettings(**{**cam.__dict__, "samples_per_px": 16384})
        conv, _, _ = orc.render(osb, conv_cam, mode=mode, rng=orc.RNG_PHILOX53, seed=99, threads=8)
        conv = conv / conv_cam.effective_spp
        cam256 = ra.CameraSettings(**{**cam.__dict__, "samples_per_px": 256})
        seeds, rmse = [11, 12, 13], []
        for sd in seeds:
            f256, _, _ = orc.render(osb, cam256, mode=mode, seed=sd, threads=8)
            rmse.append(float(np.sqrt(np.mean((f256 / cam256.effective_spp - conv) ** 2))))
        np.savez_compressed(HERE / f"{name}_oracle_64x64_converged.npz", image=conv.astype(np.float32), spp=np.array([16384]),
                            seeds=np.array(seeds), rmse_at_256spp=np.array(rmse))
        print(name, "converged; oracle rmse at 256 spp", rmse)
        np.savez_compressed(HERE / f"{name}_rays.npz", o=o, d=d, time=tm, origin_prim=frm, hit=hit, prim=prim, t=t, u=u, v=v, second_t=t2,
                            segments=np.array([cnt["segments"]]))


if __name__ == "__main__":
    main()
