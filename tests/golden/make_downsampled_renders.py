#!/usr/bin/env python3
"""Makes tests/golden/renders/disney_spheres_agx_512_ds4.png from the reference's own render
renders/disney_spheres_agx_512.png (BASELINE config 2 after AgX + sRGB, 1800x800, 8 bit): the
mean of every 4x4 pixel block, stored as 16-bit PNG channels would be overkill — values are
kept as float32 in a .npy (450x200x3, 1.0 MB) plus an 8-bit preview PNG.
Also makes sphere_{mis,mat,ref}_ds8.npy: 8x8 block means of the reference-held triplet
renders/sphere_mis.png, sphere_mat.png, sphere_ref.png (cornell_box_spheres, 800x800, 8 bit; the
author's spp and tonemapper for these three are not recorded anywhere in the reference tree).
Run in the build container (needs /root/reference); the outputs are data, not reference code."""
import os
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
src = "/root/reference/renders/disney_spheres_agx_512.png"
img = np.asarray(Image.open(src)).astype(np.float32)            # [800, 1800, 3], 0..255
ds = img.reshape(200, 4, 450, 4, 3).mean(axis=(1, 3))
np.save(os.path.join(HERE, "renders", "disney_spheres_agx_512_ds4.npy"), ds.astype(np.float16))
Image.fromarray(np.clip(ds + 0.5, 0, 255).astype(np.uint8)).save(
    os.path.join(HERE, "renders", "disney_spheres_agx_512_ds4.png"))
print("wrote", ds.shape, ds.mean(axis=(0, 1)))

for name in ("sphere_mis", "sphere_mat", "sphere_ref"):
    a = np.asarray(Image.open(f"/root/reference/renders/{name}.png")).astype(np.float32)   # [800, 800, 3]
    b = a.reshape(100, 8, 100, 8, 3).mean(axis=(1, 3))
    np.save(os.path.join(HERE, "renders", f"{name}_ds8.npy"), b.astype(np.float16))
    print("wrote", name, b.shape, b.mean(axis=(0, 1)))
