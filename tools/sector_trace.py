"""one azimuth sector (HZ_G sectors, number HZ_R) rendered 20 times, each waited for: run under
`rocprofv3 --kernel-trace --stats` (with HZ_SERIAL=1: every kernel alone) to see where a sector's time goes"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hzutil, horizonator_amd
from horizonator_amd.sharding import sector_columns, sparse_header_words, sparse_mask_stride
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
G, r = int(os.environ.get("HZ_G", "8")), int(os.environ.get("HZ_R", "1"))
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_view(-180, 180, zfar=600000.0)
c0, c1 = sector_columns(W, G, r)
h.set_sector(c0, c1)
ms = sparse_mask_stride(c1 - c0); hdr = sparse_header_words(H, ms)
sp = torch.empty(hdr + H * (c1 - c0), dtype=torch.int32, device="cuda")
for k in range(20):
    h.render_sparse(sp.data_ptr(), ms); h.sync()
