#!/usr/bin/env python3
"""Decode the reference's sample bokeh kernel into a raw texel fixture.

bokeh_imgs/example_bokeh_kernel.jpg (250x250 RGB) is input data the reference ships for
imageData::read (src/imagebokeh.h:83-140), which loads it through Arnold's AiTextureLoad as
float texels.  There is no Arnold (or image decoder) on the GPU box, so the texels are decoded
here once with PIL and committed as uint8 [250,250,3]; consumers use texel/255 as the float value
(SURVEY appendix C.18).  Run in the build container only.
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/bokeh_imgs/example_bokeh_kernel.jpg"
DST = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "example_bokeh_kernel_u8.npy")
img = np.asarray(Image.open(SRC).convert("RGB"), dtype=np.uint8)
assert img.shape == (250, 250, 3), img.shape
np.save(DST, img)
print(img.shape, img.dtype, "unique luminances:", len(np.unique(img.astype(np.int32) @ [30, 59, 11])), "->", DST)
