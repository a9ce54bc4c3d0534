"""Where the time of the Python mirror's render() goes, next to render_into() with kept buffers: setting the view, taking the
result arrays (horizonator_amd._ResultMemory), the C call.  cfg2 and cfg3, medians of 8 after 2, twice."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, ctypes as C
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
for name, R, W, H in (("cfg2", 1800, 8000, 2000), ("cfg3", 4200, 16000, 4000)):
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    h.set_view(-180, 180, zfar=600000.0)
    img = np.zeros((H, W, 3), np.uint8); rng = np.zeros((H, W), np.float32)
    for rep in range(2):
        a = []
        for _ in range(10):
            t0 = time.perf_counter(); h.render_into(img, rng); a.append(time.perf_counter() - t0)
        b = []; ph = []
        for _ in range(10):
            t0 = time.perf_counter()
            h._prepare(-180.0, 180.0, -1000.0, -1000.0, False, 100.0, 600000.0, -1.0, -1.0)
            t1 = time.perf_counter()
            i2, r2 = h._result_memory().take([((H, W, 3), np.uint8), ((H, W), np.float32)])
            t2 = time.perf_counter()
            h.render_into(i2, r2)
            t3 = time.perf_counter()
            del i2, r2
            b.append(t3 - t0); ph.append((t1 - t0, t2 - t1, t3 - t2))
        c = []
        for _ in range(10):
            t0 = time.perf_counter(); res = h.render(-180, 180, zfar=600000.0); c.append(time.perf_counter() - t0); del res
        ph = np.median(np.array(ph[2:]), axis=0) * 1e3
        print(f"{name}: render_into kept {np.median(a[2:])*1e3:.2f} ms; by hand {np.median(b[2:])*1e3:.2f} (prepare {ph[0]:.3f}, take {ph[1]:.3f}, call {ph[2]:.3f}); render() {np.median(c[2:])*1e3:.2f}", flush=True)
    h.close()
