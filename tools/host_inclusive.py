"""horizonator_render_offscreen() into caller-owned HOST memory, as the reference's API hands
results over: time per call including the device->host copies (PCIe), next to the
device-resident number bench.py reports.  Two callers: one that keeps its buffers (standalone.c),
one that gets fresh arrays from every call (the reference's Python wrapper,
horizonator-pywrap.c:234-250: PyArray_SimpleNew per render - untouched pages)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
CONFIGS = (("cfg2", 1800, 8000, 2000), ("cfg3", 4200, 16000, 4000))
for name, R, W, H in [c for c in CONFIGS if len(sys.argv) < 2 or c[0] in sys.argv[1:]]:
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    h.set_view(-180, 180, zfar=600000.0)
    img = np.zeros((H, W, 3), np.uint8); rng = np.zeros((H, W), np.float32)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); h.render_into(img, rng); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[2:]))
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); fresh = h.render(-180, 180, zfar=600000.0); ts.append(time.perf_counter() - t0); del fresh     # (freeing 448 MB is the caller's, outside the call)
    t2 = float(np.median(ts[2:]))
    import torch
    d_img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0"); d_rng = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
    h.render_device(d_img.data_ptr(), d_rng.data_ptr()); h.sync()
    same = bool(np.array_equal(img, d_img.cpu().numpy()) and np.array_equal(rng, d_rng.cpu().numpy()))
    del d_img, d_rng
    print(f"{name} [HZ_COPY_THREADS={os.environ.get('HZ_COPY_THREADS', 'default')} HZ_HOST_DENSE={os.environ.get('HZ_HOST_DENSE', '0')}] equals the device render: {same}")
    print(f"{name}: kept buffers {t*1e3:.1f} ms/call ({7*W*H/t/1e9:.1f} GB/s of results); fresh numpy arrays per call {t2*1e3:.1f} ms/call ({7*W*H/t2/1e9:.1f} GB/s)", flush=True)
    h.close()
