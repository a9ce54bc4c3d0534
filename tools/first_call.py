"""the first call of a context: into untouched pages (np.zeros) against pages touched beforehand (np.ones)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
dems = hzutil.dem_dir_for(LAT, LON, R)
for touched in (False, True, False, True):
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R)
    h.set_view(-180, 180, zfar=600000.0)
    img = (np.ones if touched else np.zeros)((H, W, 3), np.uint8); rng = (np.ones if touched else np.zeros)((H, W), np.float32)
    ts = []
    for k in range(3):
        t0 = time.perf_counter(); h.render_into(img, rng); ts.append((time.perf_counter() - t0)*1e3)
    print("buffers %s: calls %s ms" % ("touched beforehand" if touched else "untouched (np.zeros)", " ".join("%.2f" % t for t in ts)), flush=True)
    h.close(); del img, rng
