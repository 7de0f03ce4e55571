#!/usr/bin/env python3
"""Decode the XWorld2D icon set into the product's render-input asset.

Run in the build container only (needs /root/reference and Pillow):

    python tools/make_assets.py

Reads the 363 64x64 JPEG icons under games/xworld/images (the *data* the
reference renders from: xitem.cpp:33-45 `cv::imread(path, 1)` -> BGR, 3 channels)
and games/xworld/images/properties.txt (colour table, xworld_env.py:86-91) and
writes

    xworld_amd/assets/icons64.npz   icons  uint8 [n, 64, 64, 3]  BGR, like cv::imread(path, 1)
    xworld_amd/assets/icons.json    one record per icon, in the same order:
                                    path, type (goal|block|agent), subtree, name, color

Icon order is the lexicographic order of the relative path.  (The reference
orders icon variants of one name by os.walk order, which is filesystem
dependent -- xworld_env.py:247-255 -- so no particular order is pinned.)

JPEG decoding here is Pillow/libjpeg-turbo; the reference used OpenCV 3.2.0's
bundled libjpeg.  Decoders may differ by a few LSB on chroma-subsampled
images; this is the stated <= 2/255 tolerance of the rendered screen versus
(unavailable) reference pixels.  Nothing of the reference's source is copied.
"""
import json
import os
import sys

import numpy as np
from PIL import Image

REF = os.environ.get("XWORLD_REFERENCE", "/root/reference")
ROOT = os.path.join(REF, "games", "xworld", "images")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "assets")


def main():
    paths = []
    for dp, _, fs in os.walk(ROOT):
        for f in fs:
            if f.endswith(".jpg") or f.endswith(".png"):      # xworld_env.py:79-81
                paths.append(os.path.relpath(os.path.join(dp, f), ROOT))
    paths.sort()
    colors = {}
    with open(os.path.join(ROOT, "properties.txt")) as f:
        for line in f.read().splitlines():
            if line.startswith("//") or line == "":
                continue
            colors[line.split()[0]] = line.split()[1]
    icons = []
    meta = []
    for p in paths:
        im = Image.open(os.path.join(ROOT, p))
        assert im.size == (64, 64), (p, im.size)               # XItem::item_size_ (xitem.h:151)
        bgr = np.asarray(im.convert("RGB"))[:, :, ::-1]        # imread(path, 1): 3-channel BGR
        icons.append(np.ascontiguousarray(bgr))
        parts = p.split("/")
        typ = parts[0]                                         # grid_types, xworld_env.py:66
        assert typ in ("goal", "block", "agent"), p
        subtree = parts[1] if typ == "goal" else ""
        base = os.path.splitext(parts[-1])[0]
        name = "_".join(base.split("_")[:-1])                  # key = path without "_<k>", xworld_env.py:249
        meta.append({"path": p, "type": typ, "subtree": subtree, "name": name,
                     "color": colors.get(p, "na")})
    icons = np.stack(icons).astype(np.uint8)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "icons64.npz"), icons=icons)
    with open(os.path.join(OUT, "icons.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("wrote", len(meta), "icons", icons.shape, "->", OUT)


if __name__ == "__main__":
    sys.exit(main())
