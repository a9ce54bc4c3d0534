"""probe: do the strips of consecutive panoramas of one sector overlap better when they are drawn by two contexts
(two sets of streams) in turn?  ms per strip, one context vs two"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hzutil, horizonator_amd
from horizonator_amd.sharding import sector_columns, sparse_header_words, sparse_mask_stride
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
hs = [horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R) for _ in range(2)]
for G in (8, 4, 2):
    for r in (0, 1):
        c0, c1 = sector_columns(W, G, r)
        ms = sparse_mask_stride(c1 - c0); hdr = sparse_header_words(H, ms)
        bufs = []
        for h in hs:
            h.set_view(-180, 180, zfar=600000.0); h.set_sector(c0, c1)
            bufs.append(torch.empty(hdr + H * (c1 - c0), dtype=torch.int32, device="cuda"))
        res = []
        for nctx in (1, 2):
            tt = []
            for k in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for j in range(24):
                    hs[j % nctx].render_sparse(bufs[j % nctx].data_ptr(), ms)
                for h in hs: h.sync()
                tt.append((time.perf_counter() - t0) * 1e3 / 24)
            res.append(float(np.median(tt[1:])))
        print(f"G={G} r={r}: one context {res[0]:.3f} ms per strip, two in turn {res[1]:.3f}", flush=True)
