#!/usr/bin/env python3
"""The scenes the kernel was NOT tuned on: SURVEY.md 8(d)'s rough DEM,
a summit and a valley viewpoint, a 45 degree zoom, BASELINE configs[1], configs[3] (a batch of
viewpoints), configs[4] - each drawn a few times back to back on one MI355X, outputs left in HBM.

    python tools/scenes.py [--scenes a,b,...] [--steps K] [--counters] > line.json

bench.py runs the same scenes (small step counts) and puts them into its line under "scenes";
tools/history/gpu_scenes.sh swept the library's heuristics (first round's reach, one / two rounds) over
them for profiles/r3_scenes.json.

Per scene: ms per render, picoseconds per triangle of the mosaic (2 (N-1)^2 of them: what the reference
would push through its draw call, reference horizonator-lib.c:203,897), Gpix/s, and with
--counters the marching waves' own counts (one extra draw by the counting instance of the
kernel): triangles that reached the set-up stage, how many of those the early depth test
dropped, pixel centres tested.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LAT, LON = 34.4137, -117.5621          # the generic viewpoint
ZNEAR, ZFAR = 100.0, 600000.0

# name -> (R, W, H, srtm1, rough, az0, az1, viewpoint, steps)
SCENES = {
    "cfg3":          dict(R=4200, W=16000, H=4000, what="the headline: 7x7 SRTM3 tiles, 360 degrees"),
    "cfg3_rough":    dict(R=4200, W=16000, H=4000, rough=True, what="the same over SURVEY.md 8(d)'s rough DEM (+-30 m hash noise per cell)"),
    "cfg3_summit":   dict(R=4200, W=16000, H=4000, viewpoint="summit", what="viewer on the highest sample within 0.3 degrees of the window's centre"),
    "cfg3_valley":   dict(R=4200, W=16000, H=4000, viewpoint="valley", what="viewer on the lowest sample within 0.3 degrees of the window's centre"),
    "cfg3_zoom45":   dict(R=4200, W=16000, H=4000, az=(-22.5, 22.5), what="a 45 degree view at 16000x4000 (only the strips behind it are launched)"),
    "cfg3_zoom90":   dict(R=4200, W=16000, H=4000, az=(-45.0, 45.0), what="a 90 degree view at 16000x4000"),
    "cfg3_zoom180":  dict(R=4200, W=16000, H=4000, az=(-90.0, 90.0), what="a 180 degree view at 16000x4000"),
    "cfg3_zoom10":   dict(R=4200, W=16000, H=4000, az=(-5.0, 5.0), what="a 10 degree view at 16000x4000"),
    "cfg3_zoom45_summit": dict(R=4200, W=16000, H=4000, az=(-22.5, 22.5), viewpoint="summit", what="the 45 degree view from the summit viewpoint"),
    "cfg3_zoom45_valley": dict(R=4200, W=16000, H=4000, az=(-22.5, 22.5), viewpoint="valley", what="the 45 degree view from the valley viewpoint"),
    "cfg3_zoom45_rough":  dict(R=4200, W=16000, H=4000, az=(-22.5, 22.5), rough=True, what="the 45 degree view over the rough DEM"),
    "cfg3_zoom45_east":   dict(R=4200, W=16000, H=4000, az=(67.5, 112.5), what="a 45 degree view to the east"),
    "cfg3_zoom45_south":  dict(R=4200, W=16000, H=4000, az=(157.5, 202.5), what="a 45 degree view to the south"),
    "cfg3_zfar40km": dict(R=4200, W=16000, H=4000, zfar=40000.0, what="the API's default far clip (reference horizonator.h:10)"),
    "cfg2":          dict(R=1800, W=8000, H=2000, what="BASELINE configs[1]: 3x3 SRTM3 tiles, 8000x2000"),
    "cfg1":          dict(R=600, W=2000, H=500, steps=40, what="BASELINE configs[0]: one SRTM3 tile's worth, 2000x500"),
    "mid_4000":      dict(R=1800, W=4000, H=1000, steps=20, what="3x3 SRTM3 tiles, 4000x1000"),
    "cfg4_32":       dict(R=3000, W=8000, H=2000, batch=32, what="BASELINE configs[3]: viewpoints of the 16x16 lattice over 5x5 tiles, 8000x2000 BGR each, one batch"),
    "cfg5":          dict(R=19800, W=32768, H=8192, srtm1=True, steps=3, what="BASELINE configs[4]: 11x11 SRTM1 tiles (3.1 G triangles), 32768x8192"),
}
DEFAULT = ["cfg3", "cfg3_rough", "cfg3_summit", "cfg3_valley", "cfg3_zoom45", "cfg1", "cfg2", "cfg4_32", "cfg5"]


def _extreme_viewpoint(h, which, half_span_deg=0.3):
    """lat/lon of the highest / lowest sample of the context's DEM window near its centre"""
    import numpy as np
    cpd, R, t_lon, t_lat, oc_i, oc_j = h.window()
    m = h.mosaic()
    N = 2 * R
    k = int(half_span_deg * cpd)
    lo, hi = N // 2 - k, N // 2 + k
    sub = m[lo:hi, lo:hi]
    j, i = np.unravel_index(int(sub.argmax() if which == "summit" else sub.argmin()), sub.shape)
    i, j = i + lo, j + lo
    # half a cell off the sample: a viewer exactly on a grid vertex is the reference's atan(0,0) case (SURVEY.md R2 traps)
    return t_lat + (oc_j + j + 0.37) / cpd, t_lon + (oc_i + i + 0.41) / cpd, int(m[j, i])


def wave_counters(h):
    """one more draw of the current view by the counting instance of k_march: its second (or only) round"""
    import ctypes as C
    import numpy as np
    import horizonator_amd
    lib = horizonator_amd._lib.load()
    st = horizonator_amd._lib.load_selftest()   # the diagnostics entry points (include/hz_selftest.h); the context itself is the product's
    v = horizonator_amd.View()
    for k, x in h.view().items():
        setattr(v, k, x)
    cap = 8 << 20
    buf = np.zeros(cap, np.uint64)
    grid = (C.c_uint * 2)()
    if st.hz_hip_debug_wave_timing(lib.horizonator_amd_device(C.byref(h._ctx)), C.byref(v), buf.ctypes.data, cap, grid) != 0:
        return None
    a = buf[:int(grid[0]) * int(grid[1]) * 4].reshape(-1, 4)
    setup = int((a[:, 1] & 0xFFFFFFFF).sum())
    hidden = int((a[:, 3] >> 32).sum())
    return {"waves": int(a.shape[0]), "flushes": int((a[:, 1] >> 32).sum()), "triangles_set_up": setup,
            "hidden_by_early_depth_test": hidden, "early_z_kill_rate": (hidden / setup) if setup else None,
            "to_k_big": int((a[:, 2] >> 32).sum()), "pixel_centres_tested_in_the_waves": int((a[:, 3] & 0xFFFFFFFF).sum())}


def run_scene(name, steps=8, counters=False, cache=None):
    """returns the scene's record; `cache`: dict keeping contexts alive between scenes over the same window"""
    import numpy as np
    import torch
    import hzutil
    import horizonator_amd
    sc = SCENES[name]
    R, W, H = sc["R"], sc["W"], sc["H"]
    srtm1, rough = sc.get("srtm1", False), sc.get("rough", False)
    steps = sc.get("steps", steps)
    key = (R, W, H, srtm1, rough)
    init_s = 0.0
    if cache is not None and cache.get("key") == key:
        h = cache["h"]
    else:
        if cache is not None and cache.get("h") is not None:
            cache["h"].close()
            cache.clear()
            torch.cuda.empty_cache()
        dems = hzutil.dem_dir_for(LAT, LON, R, srtm1=srtm1, rough=rough)     # (synthetic tiles, written now if they are not there yet: not part of init)
        t0 = time.perf_counter()
        h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R, SRTM1=srtm1)
        init_s = time.perf_counter() - t0
        if cache is not None:
            cache.update(key=key, h=h)
    # the scenes are timed COLD, like the headline: every vertex of every render transformed in full (the renders of a
    # scene share a viewpoint, which the library's vertex cache would serve from HBM from the third on); HZ_VERTEX_CACHE=1 in
    # the environment lets it
    if os.environ.get("HZ_VERTEX_CACHE") is None:
        h.set_options(vertex_cache=0)
    az0, az1 = sc.get("az", (-180.0, 180.0))
    zfar = sc.get("zfar", ZFAR)
    rec = {"what": sc["what"], "image": [W, H], "triangles": 2 * (2 * R - 1) ** 2, "zfar_m": zfar, "init_s": init_s}
    lat, lon = LAT, LON
    if sc.get("viewpoint"):
        lat, lon, z = _extreme_viewpoint(h, sc["viewpoint"])
        rec["viewer"] = {"lat": lat, "lon": lon, "terrain_m": z}
    h.set_view(az0, az1, lat=lat, lon=lon, znear=ZNEAR, zfar=zfar)
    n = sc.get("batch", 0)
    if n:
        side = 16
        lats, lons = hzutil.viewpoint_lattice(LAT, LON, side=side)
        pick = np.linspace(0, side * side - 1, n).astype(int)           # spread over the whole lattice
        lats, lons = lats[pick], lons[pick]
        d_img = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
        h.render_batch(lats[:4], lons[:4], d_img.data_ptr(), 0)
        h.sync()
        t0 = time.perf_counter()
        h.render_batch(lats, lons, d_img.data_ptr(), 0)
        h.sync()
        dt = time.perf_counter() - t0
        per = dt / n
        rec["viewpoints"] = n
    else:
        d_img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        d_rng = torch.empty((H, W), dtype=torch.float32, device="cuda")
        for _ in range(2):
            h.render_device(d_img.data_ptr(), d_rng.data_ptr())
        h.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            h.render_device(d_img.data_ptr(), d_rng.data_ptr())
        h.sync()
        per = (time.perf_counter() - t0) / steps
        rec["terrain_fraction"] = float((d_rng[::8, ::8] >= 0).float().mean().item())
        rec["steps"] = steps
    rec["ms_per_render"] = per * 1e3
    rec["ps_per_triangle"] = per * 1e12 / rec["triangles"]
    rec["Gpix_per_s"] = W * H / per / 1e9
    if counters and not n:
        rec["counters"] = wave_counters(h)
    del d_img
    if cache is None:
        h.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default=",".join(DEFAULT))
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--counters", action="store_true")
    args = ap.parse_args()
    out, cache = {}, {}
    for name in args.scenes.split(","):
        try:
            out[name] = run_scene(name, args.steps, args.counters, cache)
        except Exception as e:          # a scene that fails (memory, a missing tile) must not lose the others
            out[name] = {"error": repr(e)}
        print(name, json.dumps(out[name])[:300], file=sys.stderr, flush=True)
    env = {k: v for k, v in os.environ.items() if k.startswith("HZ_") and k != "HZ_TEST_DEM_DIR"}
    print(json.dumps({"env": env, "scenes": out}))


if __name__ == "__main__":
    main()
