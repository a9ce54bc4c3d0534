"""what a rocprofv3 --kernel-trace run of the host path looks at: a few single calls, then a series with two in flight"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_view(-180, 180, zfar=600000.0)
img = np.zeros((H, W, 3), np.uint8); rng = np.zeros((H, W), np.float32)
img2 = np.zeros((H, W, 3), np.uint8); rng2 = np.zeros((H, W), np.float32)
for _ in range(5):
    h.render_into(img, rng)
h.sync(); time.sleep(0.01)
bufs = ((img, rng), (img2, rng2))
n = 8
h.render_begin(*bufs[0])
for k in range(1, n + 1):
    if k < n: h.render_begin(*bufs[k % 2])
    h.render_end()
h.sync()
h.close()
