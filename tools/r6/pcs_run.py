"""a few cold whole-panorama draws (cfg3), for a PC-sampling run of rocprofv3"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_options(vertex_cache=0)
h.set_view(-180, 180, zfar=600000.0)
img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda"); rng = torch.empty((H, W), dtype=torch.float32, device="cuda")
for _ in range(int(os.environ.get("PCS_DRAWS", "12"))):
    h.render_device(img.data_ptr(), rng.data_ptr())
h.sync()
h.close()
