"""Progressive rendering with resumable checkpoints.

Camera::render (camera.rs:386-401) finishes every sample of a pixel before it moves on and
writes the file once at the end.  The device adds whole strata of the full frame into a film of
sums, so a render can be cut into batches of strata: after each batch the film is a complete
image at the spp rendered so far, and (film sums, next stratum, seed) is all the state there is.
"""
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import imageio


@dataclass
class Checkpoint:
    sums: np.ndarray       # (H, W, 3) float32 film sums
    next_stratum: int      # strata [0, next_stratum) are in the sums
    n_strata: int          # sqrt_spp^2 of the render
    seed: int
    mode: int

    def save(self, path):
        tmp = Path(str(path) + ".tmp.npz")
        np.savez(tmp, sums=self.sums, meta=np.array([self.next_stratum, self.n_strata, self.seed, self.mode], dtype=np.int64))
        tmp.replace(path)  # a reader never sees a half-written file

    @staticmethod
    def load(path):
        with np.load(path, allow_pickle=False) as z:
            m = z["meta"]
            return Checkpoint(z["sums"].astype(np.float32), int(m[0]), int(m[1]), int(m[2]), int(m[3]))


def render_progressive(dev, cam_settings, mode, seed=1, batch_strata=16, checkpoint_path=None, preview_path=None,
                       on_batch=None, resume=True, **render_args):
    """Renders all strata of `cam_settings` in batches on Device `dev` (scene already uploaded).

    After every batch: optional checkpoint (npz), optional 8-bit preview (PNG, linear like
    PngWriter) scaled by the strata done so far, optional callback on_batch(done, total).
    With resume=True an existing checkpoint for the same render is continued.  Returns the
    final Checkpoint; the film stays on the device."""
    rc = cam_settings.build()
    total = cam_settings.effective_spp
    w, h = cam_settings.image_width, cam_settings.image_height
    start = 0
    if resume and checkpoint_path is not None and Path(checkpoint_path).exists():
        ck = Checkpoint.load(checkpoint_path)
        if (ck.n_strata, ck.seed, ck.mode) != (total, seed, mode) or ck.sums.shape != (h, w, 3):
            raise ValueError("checkpoint belongs to a different render")
        dev.film_upload(ck.sums)
        start = ck.next_stratum
    else:
        dev.film_resize(w, h)
    done = start
    while done < total:
        end = min(done + batch_strata, total)
        dev.render(rc, mode=mode, seed=seed, stratum_begin=done, stratum_end=end, **render_args)
        done = end
        if checkpoint_path is not None:
            Checkpoint(dev.film(1.0), done, total, seed, mode).save(checkpoint_path)
        if preview_path is not None:
            imageio.write_png(preview_path, dev.film_rgb8(1.0 / done))
        if on_batch is not None:
            on_batch(done, total)
    return Checkpoint(dev.film(1.0), done, total, seed, mode)
