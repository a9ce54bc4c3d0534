"""is a zoomed view's first round bound by atomics that miss the last-level cache?  The 45 degree view towards the east at
16000x4000 (512 MB of framebuffer) whole, and as 2 / 4 / 8 sectors of columns drawn one after the other (a sector's
framebuffer: 256 / 128 / 64 MB) - per-stage times with every kernel alone (run with HZ_SERIAL=1), summed over the sectors"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
AZ = [float(x) for x in os.environ.get("HZ_AZ", "67.5,112.5").split(",")]
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_view(AZ[0], AZ[1], znear=100.0, zfar=600000.0)
h.set_profiling(True)
for G in (1, 2, 4, 8):
    tot = {}
    for r in range(G):
        c0, c1 = r * W // G, (r + 1) * W // G
        h.set_sector(c0, c1)
        d_img = torch.empty((H, c1 - c0, 3), dtype=torch.uint8, device="cuda"); d_rng = torch.empty((H, c1 - c0), dtype=torch.float32, device="cuda")
        for k in range(3):
            h.render_device(d_img.data_ptr(), d_rng.data_ptr()); h.sync()
        for k, v in h.last_times().items():
            tot[k] = tot.get(k, 0.0) + v
    print("az", AZ, "as", G, "sectors, stage times summed:", {k: round(v, 3) for k, v in tot.items()}, flush=True)
h.close()
