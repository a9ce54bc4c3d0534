"""horizonator_render_offscreen() into caller-owned HOST memory, as the reference's API hands results over: time per call
including PCIe, next to the device-resident number bench.py reports.  Three callers: one that keeps its buffers
(standalone.c), one that gets fresh arrays from every call (the reference's Python wrapper, horizonator-pywrap.c:234-250:
PyArray_SimpleNew per render - untouched pages), one that renders a series with two sets of buffers
(horizonator_amd_render_begin / _end).  HZ_HOST_TIMES=1 adds each call's timeline on stderr; argv: configs, then
sectors=N[,N...] for a sweep of hz_options_t::host_sectors, and any number of env=NAME=VALUE[,NAME=VALUE...]: the whole
measurement once more in a context made with those variables set (the switches a context reads when it is made)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
CONFIGS = (("cfg2", 1800, 8000, 2000), ("cfg3", 4200, 16000, 4000))
names = [a for a in sys.argv[1:] if not a.startswith(("sectors=", "env="))]
sweep = [int(x) for a in sys.argv[1:] if a.startswith("sectors=") for x in a[8:].split(",")] or [0]
variants = [{}] + [dict(kv.split("=", 1) for kv in a[4:].split(",")) for a in sys.argv[1:] if a.startswith("env=")]
for name, R, W, H, env in [c + (e,) for c in CONFIGS if not names or c[0] in names for e in variants]:
    for k in set(k for e in variants for k in e): os.environ.pop(k, None)
    os.environ.update(env)
    dems = hzutil.dem_dir_for(LAT, LON, R)
    t0 = time.perf_counter()
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R)
    print(f"{name} {env or ''}: horizonator_init {time.perf_counter() - t0:.3f} s", flush=True)
    h.set_view(-180, 180, zfar=600000.0)
    import torch
    d_img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0"); d_rng = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
    h.render_device(d_img.data_ptr(), d_rng.data_ptr()); h.sync()
    want_img, want_rng = d_img.cpu().numpy(), d_rng.cpu().numpy()
    del d_img, d_rng
    for sectors in sweep:
        h.set_options(host_sectors=sectors)
        img = np.zeros((H, W, 3), np.uint8); rng = np.zeros((H, W), np.float32)
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); h.render_into(img, rng); ts.append(time.perf_counter() - t0)
        t = float(np.median(ts[2:]))
        same = bool(np.array_equal(img, want_img) and np.array_equal(rng, want_rng))
        ts2 = []
        for _ in range(9):
            t0 = time.perf_counter(); fresh = (np.empty((H, W, 3), np.uint8), np.empty((H, W), np.float32)); h.render_into(*fresh); ts2.append(time.perf_counter() - t0)
            same = same and bool(np.array_equal(fresh[0], want_img)); del fresh     # (freeing 448 MB is the caller's, outside the call)
        t2 = float(np.median(ts2[2:]))
        ts2 = []
        for _ in range(9):          # the Python mirror's render(): arrays made of the memory of results the caller dropped
            t0 = time.perf_counter(); res = h.render(-180, 180, zfar=600000.0); ts2.append(time.perf_counter() - t0)
            same = same and bool(np.array_equal(res[0], want_img)); del res
        t2py = float(np.median(ts2[2:]))
        img2 = np.zeros((H, W, 3), np.uint8); rng2 = np.zeros((H, W), np.float32)
        bufs = ((img, rng), (img2, rng2)); img[:] = 0; rng[:] = 0
        n = 14; marks = []
        h.render_begin(*bufs[0])
        for k in range(1, n + 1):
            if k < n: h.render_begin(*bufs[k % 2])
            h.render_end(); marks.append(time.perf_counter())
        t3 = float(np.median(np.diff(marks[2:])))
        same = same and bool(np.array_equal(img, want_img) and np.array_equal(rng2, want_rng) and np.array_equal(img2, want_img) and np.array_equal(rng, want_rng))
        print(f"{name} [host_sectors={sectors or 'auto'} HZ_COPY_THREADS={os.environ.get('HZ_COPY_THREADS', 'default')} HZ_HOST_DENSE={os.environ.get('HZ_HOST_DENSE', '0')}] equals the device render: {same}")
        print(f"{name}: calls in order, ms: " + " ".join(f"{x*1e3:.2f}" for x in ts))
        print(f"{name}: kept buffers {t*1e3:.2f} ms/call (min {min(ts[2:])*1e3:.2f}, max {max(ts[2:])*1e3:.2f}; {7*W*H/t/1e9:.1f} GB/s of results); fresh numpy arrays per call {t2*1e3:.2f} ms/call, Python render() {t2py*1e3:.2f}; "
              f"two in flight {t3*1e3:.2f} ms per panorama", flush=True)
        del img, rng, img2, rng2
    h.close()
