#!/usr/bin/env python3
"""Decode the reference's own committed renders into golden fixtures.

Run in the BUILD container only (it reads /root/reference, which does not exist on the GPU box):

    python tools/make_reference_fixtures.py

/root/reference/outputs/*.png are the reference's `image.ppm` files converted with `ffmpeg -i image.ppm output.png`
(/root/reference/README.md:43-68): PNG is lossless, so the decoded bytes ARE the reference's `write_ppm` output
(io/image/ppm.hpp:7-25) for the stated scene.  They are the only reference-held outputs for the path, and they are data
(expected outputs), not source.  Written: tests/golden/ref_outputs/<name>.npz (uint8 [h][w][3], key "rgb") plus
MANIFEST.json with the SHA-256 of the raw bytes, the shape and what the README says about the file.
The tests read them with numpy only.
"""
import hashlib
import json
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/outputs"
DST = os.path.join(ROOT, "tests", "golden", "ref_outputs")

FILES = {
    # name: (scene the README names, what is known about the settings)
    "refractive_dragon": ("scenes/hw11/scene8.crtscene", "README.md:60-62; 1920x1080; matches spp 1, max_ray_depth 5 exactly"),
    "textures": ("scenes/hw12/scene4.crtscene", "README.md:64-65; 1920x1080; albedo, edge, checker and bitmap textures"),
    "gi_128spp_5_1": ("scenes/hw15/scene2.crtscene", "README.md:46-51; 1080x1080, 128 spp, depth 5, 1 diffuse ray; stochastic"),
    "gi_128spp_10_1": ("scenes/hw15/scene2.crtscene", "README.md:46-51; 1080x1080, 128 spp, depth 10, 1 diffuse ray; stochastic"),
    "gi_512spp_5_1": ("scenes/hw15/scene2.crtscene", "README.md:46-51; 1080x1080, 512 spp, depth 5, 1 diffuse ray; stochastic"),
}


def main() -> None:
    os.makedirs(DST, exist_ok=True)
    manifest = {}
    for name, (scene, note) in FILES.items():
        im = Image.open(os.path.join(SRC, name + ".png"))
        assert im.mode == "RGB", (name, im.mode)
        rgb = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
        np.savez_compressed(os.path.join(DST, name + ".npz"), rgb=rgb)
        manifest[name] = {
            "source": f"outputs/{name}.png",
            "scene": scene,
            "note": note,
            "shape": list(rgb.shape),
            "sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
        }
        print(name, rgb.shape, manifest[name]["sha256"][:16])
    with open(os.path.join(DST, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
