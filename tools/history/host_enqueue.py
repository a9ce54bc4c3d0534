"""host time of one asynchronous render call (what the calling thread spends queueing a draw and its conversion),
against the time the device needs for it: full cfg3, a 1/8 sector of it as a sparse strip, cfg1"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hzutil, horizonator_amd
from horizonator_amd.sharding import sector_columns, sparse_header_words, sparse_mask_stride
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
for name, R, W, H, G in (("cfg3", 4200, 16000, 4000, 1), ("cfg3 1/8 sector, sparse strip", 4200, 16000, 4000, 8), ("cfg1", 600, 2000, 500, 1)):
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    h.set_view(-180, 180, zfar=600000.0)
    c0, c1 = sector_columns(W, G, 1 if G > 1 else 0)
    h.set_sector(c0, c1)
    SW = c1 - c0
    if G > 1:
        ms = sparse_mask_stride(SW); buf = torch.empty(sparse_header_words(H, ms) + H * SW, dtype=torch.int32, device="cuda")
        call = lambda: h.render_sparse(buf.data_ptr(), ms)
    else:
        img = torch.empty((H, SW, 3), dtype=torch.uint8, device="cuda"); rng = torch.empty((H, SW), dtype=torch.float32, device="cuda")
        call = lambda: h.render_device(img.data_ptr(), rng.data_ptr())
    for _ in range(5): call()
    h.sync()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): call()
    t1 = time.perf_counter()
    h.sync()
    t2 = time.perf_counter()
    print(f"{name}: host {1e6*(t1-t0)/n:.0f} us per call to queue it, device {1e6*(t2-t0)/n:.0f} us per render (back to back)", flush=True)
    h.close()
