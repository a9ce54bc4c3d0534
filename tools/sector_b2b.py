"""sector HZ_R of HZ_G rendered back to back as sparse strips: ms per strip (experiments with HZ_* switches)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hzutil, horizonator_amd
from horizonator_amd.sharding import sector_columns, sparse_header_words, sparse_mask_stride
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_view(-180, 180, zfar=float(os.environ.get("HZ_ZFAR", "600000")))
res = []
for G in [int(x) for x in os.environ.get("HZ_G", "8,4").split(",")]:
    for r in ([0, 1] if G > 1 else [0]):
        c0, c1 = sector_columns(W, G, r)
        h.set_sector(c0, c1)
        ms = sparse_mask_stride(c1 - c0); hdr = sparse_header_words(H, ms)
        sp = torch.empty(hdr + H * (c1 - c0), dtype=torch.int32, device="cuda")
        tt = []
        for k in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for j in range(24):
                h.render_sparse(sp.data_ptr(), ms)
            h.sync(); tt.append((time.perf_counter() - t0) * 1e3 / 24)
        res.append(f"G={G} r={r}: {np.median(tt[1:]):.3f}")
print(" | ".join(res))
