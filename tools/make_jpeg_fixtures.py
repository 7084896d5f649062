#!/usr/bin/env python3
"""Small JPEG files for the bitmap-texture decoder tests (tests/golden/jpeg/).

Run in the build container (needs PIL, which writes the files through libjpeg):  python tools/make_jpeg_fixtures.py
These are OUR inputs, not reference data: the reference holds one bitmap (scenes/hw12/textures/dragon.jpg, 4:4:4 baseline), whose
decode is pinned by outputs/textures.png.  The layouts stb_image also decodes but no reference file exercises — subsampled
chroma, greyscale, restart intervals, odd sizes — are covered here by requiring the product's C++ decoder (csrc/jpeg.cpp) and
the oracle's Python one (oracle/stb_jpeg.py), two independent restatements of stb_image, to agree on every byte, and both to
stay within a few levels of libjpeg's decode stored next to each file (<name>.libjpeg.npy): "parity unpinned" against stb itself.
"""
import io
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "tests", "golden", "jpeg")


def picture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 5.0 + seed), 128 + 100 * np.cos(y / 7.0), (x * 5 + y * 3) % 256], axis=-1)
    img += rng.normal(0, 12, img.shape)
    img[h // 3: h // 2, w // 4: w // 2] = (250, 10, 10)          # a saturated block with hard edges
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = {
    # name: (w, h, PIL save options, greyscale)
    "s444_q90": (40, 24, dict(quality=90, subsampling=0), False),
    "s422_q85": (41, 23, dict(quality=85, subsampling=1), False),
    "s420_q75": (37, 29, dict(quality=75, subsampling=2), False),
    "s420_q30_restart": (64, 48, dict(quality=30, subsampling=2, restart_marker_blocks=3), False),
    "s444_q100_restart_rows": (33, 17, dict(quality=100, subsampling=0, restart_marker_rows=1), False),
    "grey_q80": (19, 31, dict(quality=80), True),
    "one_pixel": (1, 1, dict(quality=90, subsampling=2), False),
    "s420_two_wide": (2, 9, dict(quality=60, subsampling=2), False),
    "progressive_refused": (24, 24, dict(quality=80, progressive=True), False),
}


def main():
    os.makedirs(DST, exist_ok=True)
    for k, (name, (w, h, opts, grey)) in enumerate(CASES.items()):
        img = picture(w, h, k)
        im = Image.fromarray(img).convert("L") if grey else Image.fromarray(img)
        buf = io.BytesIO()
        im.save(buf, "JPEG", **opts)
        data = buf.getvalue()
        with open(os.path.join(DST, name + ".jpg"), "wb") as f:
            f.write(data)
        dec = np.asarray(Image.open(io.BytesIO(data)))
        np.save(os.path.join(DST, name + ".libjpeg.npy"), dec if dec.ndim == 3 else dec[:, :, None])
        print(name, len(data), dec.shape)


if __name__ == "__main__":
    main()
