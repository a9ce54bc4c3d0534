#!/usr/bin/env python3
"""BASELINE.json configs[3]: a batch of 256 viewpoints over one 5x5-tile SRTM3
window, 8000x2000 each, sharded by viewpoint, BGR images gathered to rank 0.

    python tools/cfg4_batch.py [--n 256] [--zfar 600000] [--repeat 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/cfg4_batch.py

Prints one JSON line (rank 0): Mpix/s of the whole batch with the images left
in HBM, and the SHA-256 of the gathered batch (equal for every N).  Not the
bench line (bench.py measures configs[2]); a measured data point for DESIGN.md.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--zfar", type=float, default=600000.0)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--ranges", action="store_true", help="also produce the float32 range images")
    ap.add_argument("--sha", action="store_true", help="hash the gathered batch (copies it to the host)")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import hzutil
    import horizonator_amd
    from horizonator_amd.sharding import gather_viewpoints, viewpoint_slice

    R, W, H = 3000, 8000, 2000
    LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
    if rank == 0:
        hzutil.dem_dir_for(LAT, LON, R)
    if world > 1:
        dist.barrier()
    dems = hzutil.dem_dir_for(LAT, LON, R)
    os.environ["HORIZONATOR_HIP_DEVICE"] = str(local_rank)
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R)
    h.set_view(-180.0, 180.0, znear=100.0, zfar=args.zfar)

    side = int(round(args.n ** 0.5))
    lats, lons = hzutil.viewpoint_lattice(LAT, LON, side=side)
    n = side * side
    v0, v1 = viewpoint_slice(n, world, rank)
    d_img = torch.empty((v1 - v0, H, W, 3), dtype=torch.uint8, device=dev)
    d_rng = torch.empty((v1 - v0, H, W), dtype=torch.float32, device=dev) if args.ranges else None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    best = None
    batch = None
    for _ in range(args.repeat + 1):                 # first pass = warm-up
        fence()
        t0 = time.perf_counter()
        h.render_batch(lats[v0:v1], lons[v0:v1], d_img.data_ptr(), d_rng.data_ptr() if args.ranges else 0)
        h.sync()
        t_render = time.perf_counter() - t0
        batch = gather_viewpoints(d_img, n)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt, t_render], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, t_render = float(t[0]), float(t[1])
        if best is None or _ > 0 and dt < best[0]:
            best = (dt, t_render)
    dt, t_render = best
    if rank == 0:
        out = {"workload": f"cfg4: {n} viewpoints (lattice +-0.2 deg) over one 5x5-tile SRTM3 window, R={R}, "
                           f"{W}x{H} 360deg each, zfar {args.zfar:g} m",
               "n_gpus": world, "parallelism": f"viewpoints x{world}" + (" + RCCL gather of BGR8 images" if world > 1 else ""),
               "batch_s": dt, "render_s_max_rank": t_render, "ms_per_viewpoint": dt / n * 1e3,
               "value": n * W * H / dt / 1e6, "unit": "Mpix/s", "outputs": "BGR8" + (" + float32 range" if args.ranges else "")}
        if args.sha:
            hsh = hashlib.sha256()
            for v in range(n):
                hsh.update(batch[v].cpu().numpy().tobytes())
            out["batch_sha256"] = hsh.hexdigest()
        print(json.dumps(out))
    h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
