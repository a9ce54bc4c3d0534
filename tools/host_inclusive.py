"""horizonator_render_offscreen() into caller-owned HOST memory, as the
reference's API hands results over: time per call including the device->host
copies (PCIe), next to the device-resident number bench.py reports"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
for name, R, W, H in (("cfg2", 1800, 8000, 2000), ("cfg3", 4200, 16000, 4000)):
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    for _ in range(2):
        h.render(-180, 180, zfar=600000.0)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); h.render(-180, 180, zfar=600000.0); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print(f"{name}: render() to pageable host arrays {t*1e3:.1f} ms/call -> {W*H/t/1e6:.0f} Mpix/s ({7*W*H/t/1e9:.1f} GB/s of results over PCIe, incl. numpy allocation)")
    h.close()
