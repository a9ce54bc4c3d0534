"""how long does one GPU's share of a panorama take when the panorama is split
into G azimuth sectors?  (1-GPU estimate of the strong-scaling render time;
the RCCL gather is not included)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hzutil, horizonator_amd
from horizonator_amd.sharding import sector_columns
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, int(os.environ.get('HZ_H', '4000'))
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_view(-180, 180, zfar=600000.0)
h.set_profiling(True)
for G in (1, 2, 4, 8):
    worst = 0
    for r in range(G):
        c0, c1 = sector_columns(W, G, r)
        h.set_sector(c0, c1)
        img = torch.empty((H, c1-c0, 3), dtype=torch.uint8, device="cuda")
        rng = torch.empty((H, c1-c0), dtype=torch.float32, device="cuda")
        ts = []
        for k in range(6):
            h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
            ts.append(h.last_times()["total_ms"])
        if G == 8: print("   sector", r, {k: round(v, 3) for k, v in h.last_times().items()})
        worst = max(worst, float(np.median(ts[1:])))
    print(f"G={G}: slowest sector {worst:.3f} ms device time -> {W*H/worst/1e3:.0f} Mpix/s if perfectly overlapped")
