"""Fixture generator (build container only: reads /root/reference): the texels of the reference's textures/viking_room.png,
decoded with Pillow (an independent decoder -- the product's own PNG reader is compared with it in tests/test_texture.py), as
RGBA8 in tests/golden/viking_room_texels.npz.  Data, not code: the reference ships this texture and never samples it
(SURVEY.md section 0); BASELINE configs[2] ("viking_room.obj textured") runs on these texels on the GPU box, where
/root/reference does not exist.
usage: python tests/golden/make_viking_texture.py"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
src = "/root/reference/textures/viking_room.png"
img = np.asarray(Image.open(src).convert("RGBA"), dtype=np.uint8)
assert img.shape == (1024, 1024, 4), img.shape
np.savez_compressed(os.path.join(HERE, "viking_room_texels.npz"), rgba8=img)
print("wrote", img.shape, "sum", int(img.sum()))
