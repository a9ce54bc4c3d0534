#!/usr/bin/env python3
"""Would a series of renders go faster through TWO independent sets of streams, queues and framebuffers?

A context's renders follow one another on the same streams: the next panorama's first-round chain (marching kernel ->
clipper -> k_big -> coarse depth) waits for this one's, whatever else the chip has room for.  Two contexts over the
same DEM, drawn in turn from one thread, have no such order between them - if their aggregate rate beats one context's
series, a second lane inside a context would pay for the scenes whose period is that chain (the 40 km far clip, cfg2,
cfg1).  Prints ms per render: one context, two contexts in turn; and how long the host took to queue a render (one, lanes, one, ...).

    python tools/two_lanes.py [--scenes cfg3,cfg3_zfar40km,cfg2,cfg1] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="cfg3,cfg3_zfar40km,cfg2,cfg1")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--lanes", type=int, default=2)
    args = ap.parse_args()
    import torch
    import hzutil
    import horizonator_amd
    import scenes as S
    out = {}
    for name in args.scenes.split(","):
        sc = S.SCENES[name]
        R, W, H = sc["R"], sc["W"], sc["H"]
        dems = hzutil.dem_dir_for(S.LAT, S.LON, R, srtm1=sc.get("srtm1", False), rough=sc.get("rough", False))
        hs, outs = [], []
        for _ in range(args.lanes):
            h = horizonator_amd.horizonator(S.LAT, S.LON, W, H, dir_dems=dems, render_radius_cells=R)
            h.set_options(vertex_cache=0)
            az0, az1 = sc.get("az", (-180.0, 180.0))
            h.set_view(az0, az1, lat=S.LAT, lon=S.LON, znear=S.ZNEAR, zfar=sc.get("zfar", S.ZFAR))
            hs.append(h)
            outs.append((torch.empty((H, W, 3), dtype=torch.uint8, device="cuda"),
                         torch.empty((H, W), dtype=torch.float32, device="cuda")))
        steps = args.steps * (4 if name == "cfg1" else 1)

        def series(which, n):
            for k in range(2 * len(which)):
                h = hs[which[k % len(which)]]
                o = outs[which[k % len(which)]]
                h.render_device(o[0].data_ptr(), o[1].data_ptr())
            for w in which:
                hs[w].sync()
            t0 = time.perf_counter()
            for k in range(n):
                w = which[k % len(which)]
                hs[w].render_device(outs[w][0].data_ptr(), outs[w][1].data_ptr())
            queued = (time.perf_counter() - t0) / n * 1e3
            for w in which:
                hs[w].sync()
            host_ms.append(round(queued, 4))
            return (time.perf_counter() - t0) / n * 1e3

        host_ms = []
        rec = {"one": [], "lanes": []}
        for rep in range(3):
            rec["one"].append(round(series([0], steps), 4))
            rec["lanes"].append(round(series(list(range(args.lanes)), steps), 4))
        same = all(bool(torch.equal(outs[0][0], outs[k][0])) and bool(torch.equal(outs[0][1], outs[k][1])) for k in range(1, args.lanes))
        rec["same_bytes"] = same
        rec["host_ms_to_queue_a_render"] = host_ms
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
        for h in hs:
            h.close()
        del outs
        torch.cuda.empty_cache()
    print(json.dumps({"lanes": args.lanes, "scenes": out}))


if __name__ == "__main__":
    main()
