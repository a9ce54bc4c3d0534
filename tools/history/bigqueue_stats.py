"""diagnostics: what k_big gets to do on the benchmark scene - records, work items, tiles, tiles that
can hold a covered pixel, covered pixels (numpy restatement of its tiling and tile-reject test)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import hzutil, oracle
from horizonator_amd import _lib as hzlib
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
d = hzutil.dem_dir_for(LAT, LON, R)
od = oracle.Dem(LAT, LON, d, radius_cells=R)
m = od.mosaic()
AZ = [float(x) for x in os.environ.get("HZ_BQ_AZ", "-180,180").split(",")]      # (HZ_BQ_AZ=-22.5,22.5: a zoomed view)
v = od.view(LAT, LON, W, H, AZ[0], AZ[1], zfar=600000.0)
with hzutil.HipDev(m, W, H, raster=2) as dev:
    out = dev.render(v)
    print("view", AZ, "terrain pixels %.1f M of %.1f M" % ((out["ranges"] > 0).sum() / 1e6, W * H / 1e6))
    lib = dev.lib
    for which in (0, 1):
        cnt = (C.c_uint * 6)()
        MAXR = 1 << 21
        recs = np.zeros((MAXR, 10), np.int32)
        assert hzutil.hzlib.load_selftest().hz_hip_debug_bigqueue(dev.dev, which, cnt, MAXR, recs.ctypes.data) == 0
        n = min(cnt[0], MAXR)
        r = recs[:n].astype(np.int64)
        bw, bh = r[:, 2], r[:, 3]
        twl = np.where(bw > 32, 6, np.where(bw > 16, 5, np.where(bw > 8, 4, 3)))
        tw, th = 1 << twl, 64 >> twl
        tx, ty = (bw + tw - 1) >> twl, (bh + th - 1) // th
        tiles = tx * ty
        items = (tiles + 63) // 64
        print(f"set {which}: counters {list(cnt)}; records {n}, items {items.sum()}, tiles {tiles.sum()}, box pixels {int((bw*bh).sum())}")
        print("   tiles per record: p10 %d p50 %d p90 %d p99 %d max %d; records with <=4 tiles %.1f%%, <=16 %.1f%%, <=64 %.1f%%" % (
            *np.percentile(tiles, [10, 50, 90, 99, 100]), 100*(tiles <= 4).mean(), 100*(tiles <= 16).mean(), 100*(tiles <= 64).mean()))
        print("   box: width p50 %d p90 %d p99 %d; height p50 %d p90 %d p99 %d" % (*np.percentile(bw, [50, 90, 99]), *np.percentile(bh, [50, 90, 99])))
        # triangle area in pixels (snapped) vs box area: how much of a box is covered
        dx, dy = r[:, 4:7], r[:, 7:10]                          # edge vectors m -> m+1
        area = np.abs(dx[:, 0]*dy[:, 1] - dx[:, 1]*dy[:, 0]) / 2.0 / 65536.0
        print("   triangle area (unclipped) / box area: mean %.3f; sum of areas %.1f Mpx vs boxes %.1f Mpx" % (np.mean(np.minimum(area/(bw*bh), 1)), area.sum()/1e6, (bw*bh).sum()/1e6))
