"""per-stage times (HIP events; run with HZ_SERIAL=1 for each kernel alone) of some scenes of tools/scenes.py"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, numpy as np
import hzutil, horizonator_amd, scenes
for name in sys.argv[1:]:
    sc = scenes.SCENES[name]
    R, W, H = sc["R"], sc["W"], sc["H"]
    h = horizonator_amd.horizonator(scenes.LAT, scenes.LON, W, H, dir_dems=hzutil.dem_dir_for(scenes.LAT, scenes.LON, R, rough=sc.get("rough", False)), render_radius_cells=R)
    az0, az1 = sc.get("az", (-180.0, 180.0))
    lat, lon = scenes.LAT, scenes.LON
    if sc.get("viewpoint"):
        lat, lon, _ = scenes._extreme_viewpoint(h, sc["viewpoint"])
    h.set_view(az0, az1, lat=lat, lon=lon, znear=100.0, zfar=sc.get("zfar", 600000.0))
    h.set_profiling(True)
    d_img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda"); d_rng = torch.empty((H, W), dtype=torch.float32, device="cuda")
    for k in range(3):
        h.render_device(d_img.data_ptr(), d_rng.data_ptr()); h.sync()
    print(name, {k: round(v, 3) for k, v in h.last_times().items()}, flush=True)
    h.close()
