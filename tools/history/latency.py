"""one render at a time, each waited for: the latency of a single panorama (the pipelined
throughput is bench.py's business)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
for name, R, W, H in (("cfg1", 600, 2000, 500), ("cfg2", 1800, 8000, 2000), ("cfg3", 4200, 16000, 4000)):
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda"); rng = torch.empty((H, W), dtype=torch.float32, device="cuda")
    for zfar in (600000.0, 40000.0):
        h.set_view(-180, 180, zfar=zfar)
        ts = []
        for _ in range(25):
            t0 = time.perf_counter(); h.render_device(img.data_ptr(), rng.data_ptr()); h.sync(); ts.append(time.perf_counter() - t0)
        print(f"{name} zfar {zfar/1e3:.0f} km: one render waited for {np.median(ts[5:])*1e3:.3f} ms (min {min(ts[5:])*1e3:.3f})", flush=True)
    h.close()
