#!/usr/bin/env python3
"""horizonator_init() and what it is made of (HZ_INIT_TIMES=1: the library's own account on stderr), for BASELINE's
configurations:  python tools/init_times.py cfg3 cfg5   (HORIZONATOR_INGEST=host: round 1's host-side decode)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HZ_INIT_TIMES", "1")
import hzutil, horizonator_amd
CONFIGS = {"cfg1": (600, 2000, 500, False), "cfg2": (1800, 8000, 2000, False), "cfg3": (4200, 16000, 4000, False), "cfg5": (19800, 32768, 8192, True)}
for name in sys.argv[1:] or ["cfg3"]:
    R, W, H, srtm1 = CONFIGS[name]
    dems = hzutil.dem_dir_for(hzutil.VIEW_LAT, hzutil.VIEW_LON, R, srtm1=srtm1)
    for rep in range(2):
        sys.stderr.write("---- %s, init %d\n" % (name, rep)); sys.stderr.flush()
        t0 = time.perf_counter()
        h = horizonator_amd.horizonator(hzutil.VIEW_LAT, hzutil.VIEW_LON, W, H, dir_dems=dems, render_radius_cells=R, SRTM1=srtm1)
        print("%s: horizonator_init %.3f s" % (name, time.perf_counter() - t0), flush=True)
        h.close()
