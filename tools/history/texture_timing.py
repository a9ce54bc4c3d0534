"""cost of the textured resolve (row N4) on the benchmark scene: the same 16000x4000 render
with and without a caller-supplied texture (hzutil.hash_texture)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hzutil
import horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_view(-180, 180, zfar=600000.0)
h.set_profiling(True)
img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
rng = torch.empty((H, W), dtype=torch.float32, device="cuda")


def run(n=10):
    for _ in range(2):
        h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
    t0 = time.perf_counter()
    res = []
    for _ in range(n):
        h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
        res.append(h.last_times()["resolve_ms"])
    return (time.perf_counter() - t0) / n * 1e3, float(np.mean(res))


plain = run()
x0, y0, nx, ny = h.texture_layout()
h.set_texture(hzutil.hash_texture(ny * 256, nx * 256, seed=1, blocky=4))
tex = run()
print(f"texture {nx * 256}x{ny * 256} texels ({nx}x{ny} tiles)")
print(f"plain:    {plain[0]:.3f} ms per render, resolve {plain[1]:.3f} ms")
print(f"textured: {tex[0]:.3f} ms per render, resolve + shade {tex[1]:.3f} ms")
