"""slowest-sector device time at G=8 (and G=4) for equal sectors and for work-balanced layouts
(sharding.balanced_layout over sharding.azimuth_density with different floors)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hzutil, horizonator_amd
from horizonator_amd.sharding import sector_columns, balanced_layout, azimuth_density
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_view(-180, 180, zfar=600000.0)
h.set_profiling(True)
coslat = float(np.cos(np.radians(LAT)))


def measure(layout):
    ts = []
    for c0, c1 in layout:
        h.set_sector(c0, c1)
        img = torch.empty((H, c1 - c0, 3), dtype=torch.uint8, device="cuda")
        rng = torch.empty((H, c1 - c0), dtype=torch.float32, device="cuda")
        t = []
        for k in range(5):
            h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
            t.append(h.last_times()["total_ms"])
        ts.append(float(np.median(t[1:])))
    return ts


for G in (8, 4):
    eq = [sector_columns(W, G, r) for r in range(G)]
    t = measure(eq)
    print(f"G={G} equal sectors: max {max(t):.3f} mean {np.mean(t):.3f}", [round(x, 3) for x in t])
    for floor in (1.0, 0.5, 0.25, 0.1, 0.0):
        lay = balanced_layout(azimuth_density(W, -180, 180, coslat, floor=floor), G)
        t = measure(lay)
        print(f"G={G} balanced floor {floor}: max {max(t):.3f} mean {np.mean(t):.3f} widths", [c1 - c0 for c0, c1 in lay])
