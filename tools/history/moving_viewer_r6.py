"""a viewer that moves between horizonator_render_offscreen() calls: what the host spends per call (HZ_HOST_TIMES)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
img = np.zeros((H, W, 3), np.uint8); rng = np.zeros((H, W), np.float32)
h.set_view(-180, 180, zfar=600000.0)
for k in range(3): h.render_into(img, rng)
for k in range(8):
    t0 = time.perf_counter(); h.set_view(-180.0, 180.0, lat=LAT + 1e-4*(k+1), lon=LON, zfar=600000.0); t1 = time.perf_counter()
    h.render_into(img, rng); t2 = time.perf_counter()
    print("move %.3f ms, call %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3), flush=True)
h.close()
