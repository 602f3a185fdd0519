#!/usr/bin/env python3
"""Reduces two pictures the reference holds of its own output (README.md:50-51) to small fixtures.

    images/obj.png      1500x800 RGB: the reference's render of scene 2 (viking room), the size of its benchmark
                        frames -> luminance, 4x4 box mean, 8 bit: obj_lum_375x200.png
    images/obj_bvh.png  2724x1441 RGBA: a screen shot of its `-a` heat map of the same scene, scaled by the viewer
                        and cropped a little -> luminance, 4x4 box mean, 8 bit: obj_bvh_lum_681x360.png

These are outputs of the reference (data), not source.  The render depends on an HDR environment that the checkout
lacks (.MISSING_LARGE_BLOBS), so it pins what does not depend on the light: camera, projection, geometry, instance
matrix, texture coordinates, image orientation (tests/test_reference_images.py).  The heat map depends on none of
that: it is compared as a picture.   usage (in the build container): python tests/golden/reference_images/make_fixtures.py"""
import os

import numpy as np
from PIL import Image

REF = "/root/reference/images"
HERE = os.path.dirname(os.path.abspath(__file__))


def box4(lum):
    h, w = lum.shape[0] // 4 * 4, lum.shape[1] // 4 * 4
    return lum[:h, :w].reshape(h // 4, 4, w // 4, 4).mean(axis=(1, 3))


for src, dst in (("obj.png", "obj_lum_375x200.png"), ("obj_bvh.png", "obj_bvh_lum_681x360.png")):
    rgb = np.asarray(Image.open(os.path.join(REF, src)).convert("RGB")).astype(np.float64)
    small = np.round(box4(rgb.mean(axis=2))).astype(np.uint8)
    Image.fromarray(small, mode="L").save(os.path.join(HERE, dst), optimize=True)
    print(dst, small.shape, os.path.getsize(os.path.join(HERE, dst)), "bytes")
