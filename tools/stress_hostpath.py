"""one-off stress of the host path (hz_hostpath.cpp): seeded random views on contexts large enough for azimuth sectors -
whole circles, narrow and wrapped views, far clips from 2 km to 600 km, viewers that move, depth / colour extents - each rendered
three ways that must give the same bytes: into device buffers (k_resolve4), into kept host arrays (render_into: blobs, landing
area, scatter), as the Python mirror's render() (recycled result arrays); every third case as a series with two panoramas in
flight.  Prints the seeds that differ (none is the only acceptable answer).

    python tools/stress_hostpath.py 0 120"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hzutil
import horizonator_amd

LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
lo, hi = int(sys.argv[1]), int(sys.argv[2])
CONTEXTS = ((1800, 8000, 2000), (4200, 16000, 4000), (1800, 6000, 2200))     # 2 sectors, 4 sectors, 2 sectors of odd width
bad = []
for ci, (R, W, H) in enumerate(CONTEXTS):
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    d_img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    d_rng = torch.empty((H, W), dtype=torch.float32, device="cuda")
    bufs = [(np.zeros((H, W, 3), np.uint8), np.zeros((H, W), np.float32)) for _ in range(2)]
    for seed in range(lo, hi):
        if seed % len(CONTEXTS) != ci:
            continue
        rng = np.random.default_rng(77000 + seed)
        span = float(rng.choice([360.0, rng.uniform(2.0, 40.0), rng.uniform(40.0, 359.0)]))
        az0 = float(rng.uniform(-400.0, 400.0))
        frac = R / 1200.0 * 0.5
        lat, lon = LAT + float(rng.uniform(-frac, frac)), LON + float(rng.uniform(-frac, frac))
        kw = dict(znear=float(rng.choice([100.0, 1.0, 500.0])), zfar=float(rng.choice([2000.0, 40000.0, 150000.0, 600000.0])))
        if seed % 4 == 1:
            kw.update(znear_color=float(rng.uniform(10.0, 3000.0)), zfar_color=float(rng.uniform(3500.0, 30000.0)))
        h.set_view(az0, az0 + span, lat=lat, lon=lon, **kw)
        h.render_device(d_img.data_ptr(), d_rng.data_ptr())
        h.sync()
        want = (d_img.cpu().numpy(), d_rng.cpu().numpy())
        got = {}
        for b in bufs:
            b[0][:] = 1; b[1][:] = 1
        h.render_into(*bufs[0])
        got["render_into"] = bufs[0]
        res = h.render(az0, az0 + span, lat=lat, lon=lon, **kw)
        got["render"] = res
        if seed % 3 == 0:
            h.render_begin(*bufs[1])
            h.render_begin(*bufs[0])
            h.render_end()
            h.render_end()
            got["two in flight, first"] = bufs[1]
            got["two in flight, second"] = bufs[0]
        for what, (img, rngs) in got.items():
            if not (np.array_equal(img, want[0]) and np.array_equal(rngs, want[1])):
                bad.append((seed, what))
                print("DIFFERS", seed, what, (W, H), az0, span, lat, lon, kw, int((img != want[0]).any(axis=2).sum()), int((rngs != want[1]).sum()), flush=True)
        del res, got
    h.close()
    del d_img, d_rng, bufs
    torch.cuda.empty_cache()
print(f"seeds {lo}..{hi-1}: {len(bad)} differing cases")
