"""the first horizonator_render_offscreen() calls of a process, with the library's own account of them (HZ_DRAW_TIMES, HZ_HOST_TIMES)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
dems = hzutil.dem_dir_for(LAT, LON, R)
t0 = time.perf_counter()
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R)
print("init %.3f s" % (time.perf_counter() - t0), flush=True)
h.set_view(-180, 180, zfar=600000.0)
img = np.zeros((H, W, 3), np.uint8); rng = np.zeros((H, W), np.float32)
for k in range(4):
    sys.stderr.write("---- call %d\n" % k); sys.stderr.flush()
    t0 = time.perf_counter(); h.render_into(img, rng); print("call %d: %.2f ms" % (k, (time.perf_counter() - t0)*1e3), flush=True)
h.close()
