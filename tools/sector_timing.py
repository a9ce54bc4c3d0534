"""how long does one GPU's share of a panorama take when the panorama is split
into G azimuth sectors?  (1-GPU estimate of the strong-scaling render time;
the RCCL gather is not included)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hzutil, horizonator_amd
from horizonator_amd.sharding import sector_columns
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, int(os.environ.get('HZ_H', '4000'))
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_view(-180, 180, zfar=600000.0)
h.set_profiling(True)
fit = []
for G in (1, 2, 4, 8):
    worst = 0
    worst_lat = 0
    for r in range(G):
        c0, c1 = sector_columns(W, G, r)
        h.set_sector(c0, c1)
        img = torch.empty((H, c1-c0, 3), dtype=torch.uint8, device="cuda")
        rng = torch.empty((H, c1-c0), dtype=torch.float32, device="cuda")
        ts, lat = [], []
        for k in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
            lat.append((time.perf_counter() - t0) * 1e3)
            ts.append(h.last_times()["total_ms"])
        worst_lat = max(worst_lat, float(np.median(lat[1:])))
        if G == 8: print("   sector", r, {k: round(v, 3) for k, v in h.last_times().items()})
        worst = max(worst, float(np.median(ts[1:])))
        if r == 0 and G > 1:
            # what rank 0 does on top of its own sector in bench.py: draw + pack its strip, then convert
            # all G packed strips into the full-width outputs
            widest = -(-W // G)
            pk = [torch.zeros((H, widest), dtype=torch.int32, device="cuda") for _ in range(G)]
            full_img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
            full_rng = torch.empty((H, W), dtype=torch.float32, device="cuda")
            tt = []
            for k in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                h.render_packed(pk[0].data_ptr())
                for q in range(G):
                    q0, q1 = sector_columns(W, G, q)
                    h.resolve_packed(pk[q].data_ptr(), widest, q1 - q0, q0, full_img.data_ptr(), full_rng.data_ptr())
                h.sync(); tt.append((time.perf_counter() - t0) * 1e3)
            rank0 = float(np.median(tt[1:]))
            # the same with sparse strips (terrain pixels only), bench.py's default
            from horizonator_amd.sharding import sparse_header_words, sparse_mask_stride
            ms = sparse_mask_stride(widest); hdr = sparse_header_words(H, ms)
            sp = [torch.zeros(hdr + H*widest, dtype=torch.int32, device="cuda") for _ in range(G)]
            for q in range(G):
                q0, q1 = sector_columns(W, G, q)
                h.set_sector(q0, q1); h.render_sparse(sp[q].data_ptr(), ms); h.sync()
            h.set_sector(c0, c1)
            tt = []
            for k in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                h.render_sparse(sp[0].data_ptr(), ms)
                h.resolve_sparse_gathered([(sp[q].data_ptr(), sector_columns(W, G, q)[0], sector_columns(W, G, q)[1] - sector_columns(W, G, q)[0]) for q in range(G)],
                                          ms, full_img.data_ptr(), full_rng.data_ptr())
                h.sync(); tt.append((time.perf_counter() - t0) * 1e3)
            rank0s = float(np.median(tt[1:]))
            # a rank of bench.py --gather rotate: its own sector for every panorama, the conversion of all G
            # strips for every G-th
            strips = [(sp[q].data_ptr(), sector_columns(W, G, q)[0], sector_columns(W, G, q)[1] - sector_columns(W, G, q)[0]) for q in range(G)]
            tt = []
            for k in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for j in range(2 * G):
                    h.render_sparse(sp[0].data_ptr(), ms)
                    if j % G == 0:
                        h.resolve_sparse_gathered(strips, ms, full_img.data_ptr(), full_rng.data_ptr())
                h.sync(); tt.append((time.perf_counter() - t0) * 1e3 / (2 * G))
            rot = float(np.median(tt[1:]))
            # ... and its sector alone, queued back to back (what a rank does between its conversions)
            tt = []
            for k in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for j in range(16):
                    h.render_sparse(sp[0].data_ptr(), ms)
                h.sync(); tt.append((time.perf_counter() - t0) * 1e3 / 16)
            alone = float(np.median(tt[1:]))
    extra = (f"; rank 0 (own sector + conversion of all {G} strips): packed {rank0:.3f} ms, sparse {rank0s:.3f} ms wall; "
             f"a rank of --gather rotate (sector every panorama, conversion every {G}th, back to back): {rot:.3f} ms per panorama; "
             f"sector alone back to back: {alone:.3f} ms") if G > 1 else ""
    print(f"G={G}: slowest sector {worst:.3f} ms device time (sum of stages, one render waited for; its stages overlap: "
          f"host clock around that render {worst_lat:.3f} ms){extra}")
    fit.append((1.0 / G, worst, alone if G > 1 else None))
xs = np.array([f[0] for f in fit]); ys = np.array([f[1] for f in fit])
b, a = np.polyfit(xs, ys, 1)
print(f"sum of stages ~ {a:.3f} ms fixed + {b:.3f} ms * share of the panorama")
xs2 = np.array([f[0] for f in fit if f[2] is not None]); ys2 = np.array([f[2] for f in fit if f[2] is not None])
b2, a2 = np.polyfit(xs2, ys2, 1)
print(f"sector back to back ~ {a2:.3f} ms fixed + {b2:.3f} ms * share of the panorama")
