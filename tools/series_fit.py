#!/usr/bin/env python3
"""What a series of K panoramas costs as a function of K (cfg3, cold draws, outputs left in HBM): wall time between two fences for
K = 1, 2, 3, 4, 6, 8, 12, 16, 20, 32, 50 - the fixed part (the first panorama, whose rounds have nothing to overlap with, and the
last one's conversion) against the steady state's slope.  bench.py's headline is the K = 20 point."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_options(vertex_cache=0)
h.set_view(-180, 180, zfar=float(os.environ.get("HZ_ZFAR", "600000")))
img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda"); rng = torch.empty((H, W), dtype=torch.float32, device="cuda")
for _ in range(5): h.render_device(img.data_ptr(), rng.data_ptr())
h.sync()
Ks = [1, 2, 3, 4, 6, 8, 12, 16, 20, 32, 50]
res = {}
for rep in range(3):
    for K in Ks:
        h.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): h.render_device(img.data_ptr(), rng.data_ptr())
        h.sync(); torch.cuda.synchronize()
        res.setdefault(K, []).append((time.perf_counter() - t0)*1e3)
t = {K: float(np.median(v)) for K, v in res.items()}
for K in Ks: print("K = %2d: %7.3f ms = %.4f per panorama" % (K, t[K], t[K]/K))
slope = (t[50] - t[20])/30.0
print("steady state (K = 20 -> 50): %.4f ms per panorama; fixed part at K = 20: %.3f ms" % (slope, t[20] - 20*slope))
h.close()
