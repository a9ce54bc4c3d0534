"""container-only check (needs /root/reference + Mesa llvmpipe): the oracle against
the reference's shaders on llvmpipe for scenes too large to commit as fixtures"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, oracle
from oracle import glsl_run
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
SCENES = [
    ("cfg1 zfar 200km", dict(R=600, W=2000, H=500, az=(-180, 180), zfar=200000.0)),
    ("rough DEM", dict(R=300, W=1500, H=400, az=(-180, 180), zfar=60000.0, rough=True)),
    ("narrow zoom", dict(R=300, W=1200, H=900, az=(40, 52), zfar=30000.0)),
    ("partial, odd sizes", dict(R=77, W=333, H=111, az=(-123.4, 77.7), zfar=9000.0)),
    ("high viewer", dict(R=200, W=800, H=400, az=(-180, 180), zfar=50000.0, viewer_z=6000.0)),
    ("viewer on grid vertex", dict(R=64, W=512, H=128, az=(-180, 180), lat=34.0 + 500/1200.0, lon=-118.0 + 500/1200.0)),
    ("colour extents", dict(R=128, W=640, H=160, az=(-90, 90), znear=50.0, zfar=20000.0, znear_color=2000.0, zfar_color=3000.0)),
    ("cfg2 3x3 tiles 8000x2000", dict(R=1800, W=8000, H=2000, az=(-180, 180), zfar=600000.0)),
]
for name, kw in SCENES:
    kw = dict(kw)
    R, W, H = kw.pop("R"), kw.pop("W"), kw.pop("H")
    az0, az1 = kw.pop("az")
    rough = kw.pop("rough", False)
    lat, lon = kw.pop("lat", LAT), kw.pop("lon", LON)
    d = hzutil.dem_dir_for(LAT, LON, R, rough=rough)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    v = od.view(lat, lon, W, H, az0, az1, **kw)
    t0 = time.time(); g = glsl_run.render(m, v, W, H); tg = time.time() - t0
    t0 = time.time(); o = oracle.render(m, v, W, H); to = time.time() - t0
    zeq = np.array_equal(g["z24"], o["z24"]); beq = np.array_equal(g["bgr"], o["bgr"])
    nz = int((g["z24"] != o["z24"]).sum()); nb = int((g["bgr"] != o["bgr"]).any(axis=2).sum())
    print(f"{name:28s} {W}x{H} terrain {(o['z24'] != 0xFFFFFF).mean():.3f}: z24 differs on {nz} px, colour on {nb} px  (llvmpipe {tg:.1f}s, oracle {to:.1f}s)")
