#!/usr/bin/env python3
"""coarse depth for zoomed views (hz_k_hiz.h) on / off: same bytes? how long? - scenes of tools/scenes.py, one process
(the library reads its switches when a context is created)

    python tools/hiz_ab.py [scene ...]  [--set "HZ_HIZ=1 HZ_NEAR_CELLS=512" ...]
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


_set_here = []


def run(name, settings, steps):
    import torch
    import hzutil
    import horizonator_amd
    import scenes
    for k in _set_here:                 # (the switches of the setting before this one)
        os.environ.pop(k, None)
    _set_here.clear()
    for kv in settings.split():
        k, v = kv.split("=")
        os.environ[k] = v
        _set_here.append(k)
    sc = scenes.SCENES[name]
    R, W, H = sc["R"], sc["W"], sc["H"]
    dems = hzutil.dem_dir_for(scenes.LAT, scenes.LON, R, srtm1=sc.get("srtm1", False), rough=sc.get("rough", False))
    h = horizonator_amd.horizonator(scenes.LAT, scenes.LON, W, H, dir_dems=dems, render_radius_cells=R, SRTM1=sc.get("srtm1", False))
    az0, az1 = sc.get("az", (-180.0, 180.0))
    lat, lon = scenes.LAT, scenes.LON
    if sc.get("viewpoint"):
        lat, lon, _ = scenes._extreme_viewpoint(h, sc["viewpoint"])
    h.set_view(az0, az1, lat=lat, lon=lon, znear=scenes.ZNEAR, zfar=sc.get("zfar", scenes.ZFAR))
    d_img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    d_rng = torch.empty((H, W), dtype=torch.float32, device="cuda")
    for _ in range(2):
        h.render_device(d_img.data_ptr(), d_rng.data_ptr())
    h.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.render_device(d_img.data_ptr(), d_rng.data_ptr())
    h.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    t0 = time.perf_counter()
    h.render_device(d_img.data_ptr(), d_rng.data_ptr())
    h.sync()
    one = (time.perf_counter() - t0) * 1e3
    digest = hashlib.sha256(d_img.cpu().numpy().tobytes() + d_rng.cpu().numpy().tobytes()).hexdigest()[:16]
    qc = (C.c_uint * 4)()
    h._lib.hz_hip_last_queue_counts(h._lib.horizonator_amd_device(C.byref(h._ctx)), qc)
    cnt = scenes.wave_counters(h)
    h.close()
    del d_img, d_rng
    torch.cuda.empty_cache()
    return {"scene": name, "settings": settings, "ms_per_render": round(ms, 4), "one_render_waited_for_ms": round(one, 4), "sha": digest,
            "kill_rate": cnt and cnt["early_z_kill_rate"], "set_up": cnt and cnt["triangles_set_up"], "to_k_big": cnt and cnt["to_k_big"],
            "pixel_centres": cnt and cnt["pixel_centres_tested_in_the_waves"],
            "second_round_queue": {"reach": int(qc[0]), "records": int(qc[1]), "items": int(qc[2]), "next_reach_long": int(qc[3])}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=["cfg3_zoom45"])
    ap.add_argument("--set", action="append", default=None)
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    sets = a.set or ["HZ_HIZ=0", "HZ_HIZ=1"]
    for name in a.scenes:
        shas = set()
        for st in sets:
            r = run(name, st, a.steps)
            shas.add(r["sha"])
            print(json.dumps(r), flush=True)
        print(json.dumps({"scene": name, "same_bytes_under_every_setting": len(shas) == 1}), flush=True)


if __name__ == "__main__":
    main()
