"""horizonator_render_offscreen() into caller-owned HOST memory, as the reference's API hands
results over: time per call including the device->host copies (PCIe), next to the
device-resident number bench.py reports.  Two callers: one that keeps its buffers (standalone.c),
one that gets fresh arrays from every call (the reference's Python wrapper,
horizonator-pywrap.c:234-250: PyArray_SimpleNew per render - untouched pages)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
for name, R, W, H in (("cfg2", 1800, 8000, 2000), ("cfg3", 4200, 16000, 4000)):
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    h.set_view(-180, 180, zfar=600000.0)
    img = np.zeros((H, W, 3), np.uint8); rng = np.zeros((H, W), np.float32)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); h.render_into(img, rng); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[2:]))
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); fresh = h.render(-180, 180, zfar=600000.0); ts.append(time.perf_counter() - t0); del fresh     # (freeing 448 MB is the caller's, outside the call)
    t2 = float(np.median(ts[2:]))
    print(f"{name}: kept buffers {t*1e3:.1f} ms/call ({7*W*H/t/1e9:.1f} GB/s of results); fresh numpy arrays per call {t2*1e3:.1f} ms/call ({7*W*H/t2/1e9:.1f} GB/s)", flush=True)
    h.close()
