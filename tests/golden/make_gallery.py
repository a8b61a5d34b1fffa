"""Low-resolution copies of the gallery images the reference holds (2d/README.md:20-53), the only outputs of
the reference that exist: 800x600 RGBA8 PNGs rendered through its GUI with an unseeded generator and unknown
loop counts.  Each is reduced to 30x40 block means of its 8-bit RGB values / 255 (20x20-pixel blocks) and stored in
gallery_lowres.npz beside the name of the world that produced it (2d/src/worlds.rs).  Run in the build container,
where /root/reference exists:  python tests/golden/make_gallery.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/2d"
# world constructor (2d/src/worlds.rs) -> the gallery image it produced (2d/README.md)
IMAGES = {
    "sample_world": "sample_scene.png",
    "cornell_box": "diffration_dispersion.png",
    "cornell_box_absorption": "absorption.png",
    "spectrum_world": "visible_spectrum2.png",
}


def block_means(img, rows=30, cols=40):
    h, w = img.shape[:2]
    return img[: h // rows * rows, : w // cols * cols].reshape(rows, h // rows, cols, w // cols, -1).mean(axis=(1, 3))


def main():
    out = {}
    for world, png in IMAGES.items():
        img = np.asarray(Image.open(os.path.join(REF, png)).convert("RGB"), np.float32) / 255.0
        assert img.shape == (600, 800, 3), img.shape
        out[world] = block_means(img).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "gallery_lowres.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
