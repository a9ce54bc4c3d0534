"""BASELINE config 5 at full size: 11x11 SRTM1 tiles (39600^2 samples, 3.1 G
triangles), 32768x8192 panorama.  Far beyond what the reference can load (tile
limit, 16-bit vertex coordinates, > 2^31 indices), so: size-independent
properties only - run-to-run identical, sectors tile the panorama, both GPU
rasterisers agree on a sector, primitive ids beyond 2^31 survive."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd

os.system("free -g | head -2; df -h /tmp | tail -1")
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 19800, 32768, 8192
t0 = time.time()
dems = hzutil.dem_dir_for(LAT, LON, R, srtm1=True)
print("tiles ready in %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R, SRTM1=True)
print("init %.1f s, Ntriangles (saturated int) %d, tiles %s" % (time.time() - t0, h.Ntriangles, list(h._ctx.dems.Ndems_ij)), flush=True)
h.set_profiling(True)
h.set_view(-180, 180, zfar=600000.0)
import torch
c0, c1 = 12000, 14048
h.set_sector(c0, c1)
SW = c1 - c0
img = torch.empty((H, SW, 3), dtype=torch.uint8, device="cuda"); rng = torch.empty((H, SW), dtype=torch.float32, device="cuda")
idx = torch.empty((H, SW), dtype=torch.int32, device="cuda")
h.render_device(img.data_ptr(), rng.data_ptr(), idx.data_ptr()); h.sync()
a = (img.cpu().numpy(), rng.cpu().numpy(), idx.cpu().numpy())
print("sector times", h.last_times(), flush=True)
h.set_raster(1)
h.render_device(img.data_ptr(), rng.data_ptr(), idx.data_ptr()); h.sync()
b = (img.cpu().numpy(), rng.cpu().numpy(), idx.cpu().numpy())
print("scatter sector times", h.last_times(), flush=True)
for x, y, n in zip(a, b, ("bgr", "ranges", "index")):
    assert np.array_equal(x, y), n
print("march == scatter on the sector; terrain fraction %.3f; max |id| as uint32 %d" % ((a[2] != -1).mean(), a[2].view(np.uint32)[a[2] != -1].max()))
h.set_raster(0)
h.set_sector(0, W)
del img, rng, idx
img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda"); rng = torch.empty((H, W), dtype=torch.float32, device="cuda")
idx = torch.empty((H, W), dtype=torch.int32, device="cuda")
ts = []
for k in range(3):
    h.render_device(img.data_ptr(), rng.data_ptr(), idx.data_ptr()); h.sync(); ts.append(h.last_times())
print("full render times", ts[-1], "-> %.1f Mpix/s" % (W*H/ts[-1]["total_ms"]/1e3), flush=True)
full_idx = idx[:, c0:c1].cpu().numpy()
assert np.array_equal(full_idx, a[2]), "sector != columns of the full render"
assert np.array_equal(rng[:, c0:c1].cpu().numpy(), a[1])
i1 = idx.clone()
h.render_device(img.data_ptr(), rng.data_ptr(), idx.data_ptr()); h.sync()
assert torch.equal(i1, idx), "run to run"
print("cfg5 OK")
