"""diagnostics: distribution of k_march wave durations (HZ_WAVE_TIMING)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = [int(x) for x in os.environ.get("HZ_WT_CFG", "4200,16000,4000").split(",")]     # (cfg2: 1800,8000,2000; cfg1: 600,2000,500)
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
ZFAR = float(os.environ.get("HZ_WT_ZFAR", "600000"))
AZ = [float(x) for x in os.environ.get("HZ_WT_AZ", "-180,180").split(",")]
h.set_view(AZ[0], AZ[1], zfar=ZFAR)
import torch
SW = W
if os.environ.get("HZ_WT_SECTOR"):                  # "G,r": the waves of sector r of G (one GPU's share of the panorama)
    from horizonator_amd.sharding import sector_columns
    G, r = [int(x) for x in os.environ["HZ_WT_SECTOR"].split(",")]
    c0, c1 = sector_columns(W, G, r); h.set_sector(c0, c1); SW = c1 - c0
    print("sector", r, "of", G, "columns", c0, c1)
img = torch.empty((H, SW, 3), dtype=torch.uint8, device="cuda"); rng = torch.empty((H, SW), dtype=torch.float32, device="cuda")
for _ in range(2):
    h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
import ctypes as C
lib = horizonator_amd._lib.load()
st = horizonator_amd._lib.load_selftest()     # the diagnostics entry points (include/hz_selftest.h); the context itself is the product's
v = horizonator_amd.View()
for k, x in h.view().items():
    setattr(v, k, x)
cap = 4 * 4 * 1024 * 1024
buf = np.zeros(cap, np.uint64)
grid = (C.c_uint * 2)()
assert st.hz_hip_debug_wave_timing(lib.horizonator_amd_device(C.byref(h._ctx)), C.byref(v), buf.ctypes.data, cap, grid) == 0
gx, gy = int(grid[0]), int(grid[1])
a = buf[:gx * gy * 4].reshape(gy, gx, 4)
t = a[:, :, 0].astype(np.float64) / 2400.0          # shader clock ~2.4 GHz -> us
flushes = (a[:, :, 1] >> 32).astype(np.int64); tris = (a[:, :, 1] & 0xFFFFFFFF).astype(np.int64)
big = (a[:, :, 2] >> 32).astype(np.int64); mid = (a[:, :, 2] & 0xFFFFFFFF).astype(np.int64)
items = (a[:, :, 3] & 0xFFFFFFFF).astype(np.int64); hidden = (a[:, :, 3] >> 32).astype(np.int64)
print("zfar", ZFAR, "grid", gx, gy, "waves", gx*gy)
if os.environ.get("HZ_WT_SAVE"):                    # the raw counters, for tools/wave_schedule.py
    np.save(os.environ["HZ_WT_SAVE"], a)
busy = t > 3.0
print("waves longer than 3 us: %d, their sum %.1f ms; the others: sum %.1f ms, median %.2f us" % (busy.sum(), t[busy].sum()/1e3, t[~busy].sum()/1e3, np.median(t[~busy]) if (~busy).any() else 0))
q = np.percentile(t, [50, 90, 99, 99.9, 100])
print("wave duration us: p50 %.1f p90 %.1f p99 %.1f p99.9 %.1f max %.1f; sum %.1f ms = %.1f us on each of 4096 slots" % (*q, t.sum()/1e3, t.sum()/4096))
for y in range(0, gy, max(1, gy//24)):
    print("   seg %4d: waves' us median %.1f max %.1f sum %.0f" % (y, np.median(t[y]), t[y].max(), t[y].sum()))
print("totals: flushes %d tris %d big %d mid %d inline items %d hidden by early-Z %d" % (flushes.sum(), tris.sum(), big.sum(), mid.sum(), items.sum(), hidden.sum()))
j, i = np.unravel_index(np.argsort(t.ravel())[-12:], t.shape)
for y, x in zip(j[::-1], i[::-1]):
    print("   seg %d strip %d: %.1f us flushes %d tris %d big %d mid %d items %d" % (y, x, t[y, x], flushes[y, x], tris[y, x], big[y, x], mid[y, x], items[y, x]))
# regress time on counters
X = np.stack([np.ones(t.size), flushes.ravel(), tris.ravel(), big.ravel(), mid.ravel(), items.ravel()], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(X, t.ravel(), rcond=None)
print("least squares us: const %.2f per flush %.3f per tri %.4f per big %.3f per mid %.3f per item %.5f" % tuple(coef))
