"""one-off stress: seeded random configurations beyond the 64 of the test suite (hzutil.random_view_case: narrow, wide, wrapped and
exactly-360 views, sectors, odd sizes, rough DEM, colour extents, viewer heights), plus larger images with two-round draws, the
HIP path (both rasterisers) against the oracle on every output; prints the seeds that differ (none is the only acceptable answer)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, oracle
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    c = hzutil.random_view_case(seed)
    R, W, H = c["R"], c["W"], c["H"]
    if seed % 4 == 3:                  # larger images: two rounds, early depth test, work lists with more segments
        W, H = W * 6, H * 6
        c["c0"], c["c1"] = c["c0"] * 6, c["c1"] * 6
        os.environ["HZ_TWO_PASS"] = "1"
    else:
        os.environ.pop("HZ_TWO_PASS", None)
    if os.environ.get("STRESS_HIZ"):   # every draw in two rounds with coarse depth (hz_k_hiz.h), short first rounds: large boxes in the second
        os.environ["HZ_TWO_PASS"] = "1"; os.environ["HZ_HIZ"] = "1"; os.environ["HZ_NEAR_CELLS"] = str((8, 16, 40)[seed % 3])
    d = hzutil.dem_dir_for(LAT, LON, R, rough=c["rough"])
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    v = od.view(c["lat"], c["lon"], W, H, c["az0"], c["az1"], **c["kw"])
    orc = oracle.render(m, v, W, H, c["c0"], c["c1"])
    for raster in (2, 1):
        hip = hzutil.hip_render(m, v, W, H, col0=c["c0"], col1=c["c1"], raster=raster)
        try:
            hzutil.assert_same_render(hip, orc, f"seed {seed} raster {raster}")
        except AssertionError as e:
            bad.append((seed, raster, str(e)[:200]))
            print("DIFFERS", seed, raster, str(e)[:300], c, flush=True)
print(f"seeds {lo}..{hi-1}: {len(bad)} differing (of {2*(hi-lo)} comparisons)")
