#!/usr/bin/env python3
"""bench.py - rendered-panorama throughput of the HIP render path on MI355X.

One step = one full pass of the hot path over one panorama: clear + vertex
transform + triangle emission + depth-buffered rasterisation + resolve to BGR8
and float32 ranges, with the DEM mosaic already resident in HBM and the outputs
left in HBM (device buffers).  On N > 1 GPUs the panorama is split into N
azimuth sectors (image-column ranges), one per rank, and the strips are gathered
to rank 0 over RCCL inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3]

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for what each field
means and how `roofline` is derived.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# SURVEY.md section 8(d) / BASELINE.md section 3
CONFIGS = {
    "cfg1": dict(R=600, W=2000, H=500, tiles="1 SRTM3 tile window (2x2 tiles touched)"),
    "cfg2": dict(R=1800, W=8000, H=2000, tiles="3x3 SRTM3 tile mosaic"),
    "cfg3": dict(R=4200, W=16000, H=4000, tiles="7x7 SRTM3 tile mosaic"),
    # BASELINE configs[4]: beyond what the reference can load; meant for --gpus 2..8 (a sparse strip holds up to 16384
    # columns; one GPU renders it in tools/scenes.py)
    "cfg5": dict(R=19800, W=32768, H=8192, srtm1=True, tiles="11x11 SRTM1 tile mosaic"),
}
LAT, LON = 34.4137, -117.5621
ZNEAR = 100.0
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--zfar", type=float, default=600000.0,
                    help="far clip range in m; 600 km keeps every triangle of the mosaic live")
    ap.add_argument("--raster", type=int, default=0, help="HZ_RASTER_* (0 auto, 1 scatter, 2 march)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (zfar = 40 km, the scenes)")
    ap.add_argument("--no-host", action="store_true", help="skip the host-inclusive measurement (results into host memory)")
    ap.add_argument("--no-scenes", action="store_true",
                    help="skip the scenes the kernel was not tuned on (tools/scenes.py: rough DEM, summit, valley, 45 degree zoom, "
                         "configs[1], configs[3], configs[4]; a few renders each)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="diagnostics: gloo moves the strips through host memory (lets two ranks share one GPU "
                         "to exercise the N>1 loop where only one GPU exists); the driver's runs use nccl = RCCL")
    ap.add_argument("--same-gpu", action="store_true", help="diagnostics: every rank uses GPU 0 (with --backend gloo)")
    ap.add_argument("--exchange-anyway", action="store_true",
                    help="diagnostics, 1 GPU: run the N > 1 loop (sparse strip, device-side stream ordering, exchange, "
                         "conversion of the 'gathered' strips) with the one rank there is")
    ap.add_argument("--gather", default="rotate", choices=["rotate", "root0"],
                    help="N > 1: which rank gathers and converts a panorama's strips - rotate: panorama k goes to rank "
                         "k mod N (every rank converts 1/N of the panoramas; the outputs stay on the rank that assembled "
                         "them); root0: always rank 0, which then draws a narrower sector")
    ap.add_argument("--loop", default="python", choices=["python", "c"],
                    help="N > 1 (RCCL, sparse strips): who drives the series - python: this file's loop over torch.distributed "
                         "(what has run on one GPU with two to four gloo ranks); c: horizonator_rccl_render_series, the same steps as one "
                         "C call over an RCCL communicator of the library's own (include/horizonator_rccl.h; on one GPU it has run "
                         "with the one rank there is)")
    ap.add_argument("--wire", default="sparse", choices=["sparse", "packed"],
                    help="N > 1: what a rank sends to rank 0 - sparse: terrain pixels only + mask (default); "
                         "packed: every pixel, 4 bytes")
    return ap.parse_args()


def main():
    args = parse()
    # never a hang: whatever blocks (a collective that never completes, a stream behind it), the process says where and
    # leaves with a non-zero code - long before the driver's own limit
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("BENCH_DEADLINE_S", "1500")), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: start the N ranks ourselves (as a child process -
            # nothing in this process has touched the GPU yet) and hand its JSON line through
            import socket
            import subprocess
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            sys.exit(subprocess.call(cmd, env=env))
        args.gpus = world

    if args.config == "cfg5":
        # 3.1 G triangles: the CPU oracle needs half a minute per render, the results are 1.9 GB - the side measurements are cfg3's
        args.no_cpu_baseline = args.no_host = args.no_scenes = True

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import datetime
    # a collective that does not complete within two minutes is an error with RCCL's text, not a wait for the driver's
    # half-hour limit: the watchdog of the process group then takes this process down (non-zero), torchrun the others
    pg_timeout = datetime.timedelta(seconds=float(os.environ.get("BENCH_COLLECTIVE_TIMEOUT_S", "120")))
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only does dmabuf IPC
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group("gloo", timeout=pg_timeout)
    elif args.exchange_anyway and args.backend == "nccl":
        # the one rank there is, as an RCCL process group of its own: the strips then travel through dist.gather /
        # dist.all_reduce like those of N ranks do (the same calls, the same stream ordering), not through a copy
        import socket
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for attempt in range(4):
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            try:
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev, timeout=pg_timeout)
                break
            except Exception as e:      # (the port found free a moment ago was taken in between: DistNetworkError, EADDRINUSE)
                if attempt == 3 or "EADDRINUSE" not in str(e):
                    raise

    import __graft_entry__ as entry
    if rank == 0:
        entry.build(quiet=True)
    if world > 1:
        dist.barrier()
    import hzutil
    import horizonator_amd
    from horizonator_amd.sharding import (StripExchange, agree_on_capacity, broadcast_dem, gather_strips_async, gatherer_weights,
                                          sector_columns, sparse_header_words, sparse_mask_stride)

    cfg = CONFIGS[args.config]
    R, W, H = cfg["R"], cfg["W"], cfg["H"]
    N = 2 * R

    # synthetic SRTM tiles (tools/demgen.c).  Rank 0 generates, reads and decodes them; the other
    # ranks receive the int16 mosaic by broadcast (RCCL) and never touch a tile.
    os.environ["HORIZONATOR_HIP_DEVICE"] = str(local_rank)
    t0 = time.perf_counter()
    h = None
    if rank == 0:
        dems = hzutil.dem_dir_for(LAT, LON, R, srtm1=cfg.get("srtm1", False))
        t0 = time.perf_counter()
        h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R, SRTM1=cfg.get("srtm1", False))
    if world > 1:
        window, mosaic = broadcast_dem(h.window() if rank == 0 else None, h.mosaic() if rank == 0 else None,
                                       device=dev if args.backend == "nccl" else None)
        if rank != 0:
            h = horizonator_amd.horizonator.from_mosaic(LAT, LON, W, H, window, mosaic)
        del mosaic
    init_s = time.perf_counter() - t0
    h.set_raster(args.raster)
    h.set_profiling(False)      # (timed() switches the HIP events around the kernels on for the last of its K renders)
    from horizonator_amd.sharding import azimuth_density, balanced_layout

    # N = 1: draw + readback conversion into BGR8 / float32 range, both left in HBM.
    # N > 1: every rank draws its sector and ships it without the sky: the terrain pixels as one
    # word each (z24<<8 | red8) plus a mask - the gather is what xGMI limits; rank 0 runs the
    # readback conversion on what arrives, into the full-width outputs.  Two sets of strip buffers:
    # while RCCL moves the strips of panorama k, panorama k+1 is already being drawn into the other.
    #
    # Who draws which columns.  Equal azimuth spans are not equal work (the DEM window is square in
    # cells, cells are not square in metres, corners are further away than edges): the layout
    # gives every rank the same share of the work behind the columns - first from the geometry
    # (sharding.azimuth_density), then from what the ranks measure on this scene.  Rank 0 draws a
    # bit less under --gather root0: it also converts the gathered strips (0.27 ms for 64 Mpix, on the
    # library's conversion stream beside its own draw, which that slows by about 0.08 ms; a sector's strips
    # back to back cost about 0.10 + 0.74*share ms: profiles/r4_sector_timing.txt).
    multi = world > 1 or args.exchange_anyway
    NBUF = 2 if multi else 1
    sparse = args.wire == "sparse"
    cdev = dev if args.backend == "nccl" else torch.device("cpu")       # where the collectives' tensors live
    S = {}                                                               # the current layout and its buffers

    def drop_series():
        if S.get("rs") is not None:
            S["rs"].close()
        S["rs"] = None

    def set_gather(mode):
        drop_series()
        S["rotate"] = mode == "rotate"
        S["weights"] = gatherer_weights(world, 0.74, 0.08) if world > 1 and mode != "rotate" else None
    set_gather(args.gather)

    def apply_layout(layout):
        S["layout"] = layout
        S["col0"], S["col1"] = layout[rank]
        S["SW"] = S["col1"] - S["col0"]
        S["SW_max"] = max(c1 - c0 for c0, c1 in layout)
        S["MSTRIDE"] = sparse_mask_stride(S["SW_max"])
        S["HDR"] = sparse_header_words(H, S["MSTRIDE"])
        S["ex"] = None
        drop_series()
        if S["SW"] > 0:
            h.set_sector(S["col0"], S["col1"])
        if multi:
            if sparse:
                # a sparse strip: header + one word per TERRAIN pixel; room for the worst case (no sky at all)
                S["FULL"] = S["HDR"] + H * S["SW_max"]
                # (torch.empty: the library zeroes a strip's count word itself, on its own stream; a fill on
                # torch's stream would be unordered against the render that writes the buffer)
                S["d_pk"] = [torch.empty(S["FULL"], dtype=torch.int32, device=dev) for _ in range(NBUF)]
            else:
                S["d_pk"] = [torch.empty((H, S["SW"]), dtype=torch.int32, device=dev) for _ in range(NBUF)]

    if not multi:
        apply_layout([(0, W)])
    else:
        cos_lat = float(np.cos(np.radians(LAT)))
        apply_layout(balanced_layout(azimuth_density(W, -180.0, 180.0, cos_lat, floor=0.1), world, S["weights"]))
    if True:                                # (every rank: under --gather rotate each assembles panoramas, and both modes run)
        d_img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        d_rng = torch.empty((H, W), dtype=torch.float32, device=dev)
    pending = [None] * NBUF
    state = {"k": 0, "wire_words": 0, "keep": None, "converted": 0}
    on_gpu = args.backend == "nccl"

    def my_stream():
        return torch.cuda.current_stream().cuda_stream

    def rebalance(rounds=2, probes=3):
        """N > 1, before the warm-up: every rank times its own sector on this scene; columns are then
        re-dealt so that the measured work per rank is equal (times rank 0's weight).  All ranks
        compute the same layout from the same all-gathered numbers."""
        for _ in range(rounds):
            t = 0.0
            if S["SW"] > 0:
                ts = []
                for _ in range(probes):
                    # eight strips queued back to back, as in the timed loop: a single render that is
                    # waited for mostly measures latencies that the loop hides
                    h.sync()
                    t0 = time.perf_counter()
                    for _ in range(8):
                        if sparse:
                            h.render_sparse(S["d_pk"][0].data_ptr(), S["MSTRIDE"])
                        else:
                            h.render_packed(S["d_pk"][0].data_ptr())
                    h.sync()
                    ts.append((time.perf_counter() - t0) / 8)
                t = float(np.median(ts[1:]))
            mine = torch.tensor([t], dtype=torch.float64, device=cdev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            times = [float(x.item()) for x in every]
            density = np.zeros(W)
            ok = True
            for (c0, c1), tr in zip(S["layout"], times):
                if c1 > c0:
                    if not (tr > 0.0):
                        ok = False
                    density[c0:c1] = tr / (c1 - c0)
            if not ok:
                return
            # a measurement that is off by more than 2x from the typical column is not believed
            typical = float(np.median(density[density > 0.0]))
            density = np.where(density > 0.0, np.clip(density, 0.5 * typical, 2.0 * typical), 0.0)
            # columns nobody drew (a rank of weight 0): as costly as the average drawn column
            density[density == 0.0] = density[density > 0.0].mean()
            apply_layout(balanced_layout(density, world, S["weights"]))

    def exchange():
        """N > 1, sparse strips: the ranks agree ONCE per layout on how many words a strip sends
        (the longest strip of this scene + 10 %); from then on a panorama's gather needs nothing from
        the host - no length read back, no all_reduce (sharding.StripExchange)"""
        if S["ex"] is None:
            words = S["HDR"]
            if S["SW"] > 0:
                h.render_sparse(S["d_pk"][0].data_ptr(), S["MSTRIDE"])
                h.sync()
                words += int(S["d_pk"][0][0].item())
            cap = agree_on_capacity(words, S["HDR"], S["FULL"], cdev)
            S["ex"] = StripExchange(cap, S["FULL"], S["HDR"], cdev, nslots=NBUF, any_dst=S["rotate"],
                                    collectives_even_alone=args.exchange_anyway and on_gpu)
            S["sent"] = [None] * NBUF
            state["wire_words"] = S["ex"].cap
        return S["ex"]

    c_loop = args.loop == "c"
    if c_loop and not (multi and sparse):
        raise SystemExit("--loop c drives the exchange of sparse strips: N > 1 (or --exchange-anyway), --wire sparse")
    c_transport = "rccl" if args.backend == "nccl" else "gloo"     # (gloo: diagnostics - ranks that share a GPU, through host memory)

    def c_series():
        """--loop c: the communicator, strip buffers and bins of horizonator_rccl_render_series for this layout and scene;
        the ranks agree once on the words a strip sends, as exchange() does"""
        if S.get("rs") is None:
            words = S["HDR"]
            if S["SW"] > 0:
                h.render_sparse(S["d_pk"][0].data_ptr(), S["MSTRIDE"])
                h.sync()
                words += int(S["d_pk"][0][0].item())
            from horizonator_amd.sharding import RcclSeries
            cap = agree_on_capacity(words, S["HDR"], S["FULL"], cdev)
            S["rs"] = RcclSeries(h, S["layout"], H, cap, d_img.data_ptr(), d_rng.data_ptr(), S["rotate"], dev, nslots=NBUF, transport=c_transport)
            state["wire_words"] = S["rs"].words
        return S["rs"]

    def c_run(n, check=False):
        """n panoramas through the C loop; check: every rank's strip fit the agreed words (else: once more with room for the worst case)"""
        rs = c_series()
        first = rs.next
        fit = rs.run(n, check_fit=check)
        if check:
            t = torch.tensor([int(fit)], dtype=torch.int64, device=cdev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if not int(t.item()):
                drop_series()
                from horizonator_amd.sharding import RcclSeries
                S["rs"] = RcclSeries(h, S["layout"], H, S["FULL"], d_img.data_ptr(), d_rng.data_ptr(), S["rotate"], dev, nslots=NBUF, transport=c_transport)
                state["wire_words"] = S["rs"].words
                S["rs"].run(n)
        state["converted"] += sum(1 for i in range(first, first + n) if (i % world if S["rotate"] else 0) == rank)

    def convert(bins):
        """rank 0: the strips of one panorama -> the full-width outputs, on the library's conversion
        stream, behind their arrival (a device-side wait) and beside the draw that follows"""
        if on_gpu:
            h.waits_for_stream(my_stream())
        else:
            bins = [t.to(dev) for t in bins]
        h.resolve_sparse_gathered([(t.data_ptr(), c0, c1 - c0) for t, (c0, c1) in zip(bins, S["layout"])],
                                  S["MSTRIDE"], d_img.data_ptr(), d_rng.data_ptr())
        if not on_gpu:
            h.sync()                                    # the uploaded copies go away with this frame
        state["converted"] += 1

    def finish(slot):
        """complete the exchange that still reads buffer set `slot`; rank 0: turn the strips
        into the panorama"""
        if pending[slot] is None:
            return
        if sparse:
            ex = S["ex"]
            bins, overflow = ex.complete(slot)
            if overflow:                                # a strip outgrew the agreed capacity: once more, with more room
                bins = ex.grow(slot, S["sent"][slot])
                state["wire_words"] = ex.cap
            if bins is not None:
                convert(bins)
            pending[slot] = None
            return
        parts = pending[slot].parts()
        torch.cuda.current_stream().synchronize()       # the strips have arrived (RCCL's stream -> host)
        if parts is not None:
            if args.backend == "gloo":
                parts = [(t.to(dev), c0, n) for t, c0, n in parts]
            h.resolve_gathered(parts, d_img.data_ptr(), d_rng.data_ptr())
            h.sync()                                     # ... before the strips are released
            state["converted"] += 1
        pending[slot] = None

    host_us = {"finish": 0.0, "render": 0.0, "post": 0.0, "n": 0}    # diagnostics (BENCH_HOST_TIMES=1): host time per part of a step

    def step():
        if os.environ.get("BENCH_HOST_TIMES") and multi and sparse:
            t0 = time.perf_counter(); slot = state["k"] % NBUF; finish(slot); t1 = time.perf_counter()
            host_us["finish"] += (t1 - t0) * 1e6
        slot = state["k"] % NBUF
        dst = state["k"] % world if S["rotate"] else 0      # the rank that gathers and converts this panorama
        state["k"] += 1
        if not multi:
            # no wait in between: the library overlaps the readback conversion and the clear of
            # panorama k (its second stream) with the rasterisation of panorama k+1
            h.render_device(d_img.data_ptr(), d_rng.data_ptr())
            return
        finish(slot)
        d_pk = S["d_pk"]
        if sparse:
            ex = exchange()
            t0 = time.perf_counter()
            if on_gpu:
                # the strip buffer of this slot was last read by the exchange of two panoramas ago, on torch's stream
                # (finish() made that stream wait for it): the conversion that refills it is ordered behind that
                h.waits_for_stream(my_stream())
            if S["SW"] > 0:
                h.render_sparse(d_pk[slot].data_ptr(), S["MSTRIDE"])
            if on_gpu:
                # the gather runs behind the strip's conversion on the device; the host goes on to the next panorama
                h.stream_waits_for_outputs(my_stream())
                send = d_pk[slot]
            else:
                h.sync()
                send = d_pk[slot].cpu()                 # gloo (diagnostics): through host memory
            S["sent"][slot] = send
            t1 = time.perf_counter()
            ex.post(slot, send, dst=dst)
            pending[slot] = True
            host_us["render"] += (t1 - t0) * 1e6; host_us["post"] += (time.perf_counter() - t1) * 1e6; host_us["n"] += 1
            return
        if S["SW"] > 0:
            h.render_packed(d_pk[slot].data_ptr())
            h.sync()
        # the one exchange of the path: strips -> rank 0 over RCCL/xGMI
        pending[slot] = gather_strips_async(d_pk[slot] if args.backend == "nccl" else d_pk[slot].cpu(), W,
                                            dst=dst, layout=S["layout"])

    def drain():
        for slot in range(NBUF):
            finish(slot)
        h.sync()
        state["keep"] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def verify():
        """N > 1: the panorama a rank assembled from the gathered strips must be, byte for byte,
        what one GPU renders on its own.  Every rank that converted a panorama checks its last one."""
        if not multi:
            return None
        ok = 1
        if state["converted"] > 0:
            got_img, got_rng = d_img.clone(), d_rng.clone()
            h.set_sector(0, W)
            one_img = torch.empty_like(d_img)
            one_rng = torch.empty_like(d_rng)
            h.render_device(one_img.data_ptr(), one_rng.data_ptr())
            h.sync()
            if S["SW"] > 0:
                h.set_sector(S["col0"], S["col1"])
            ok = int(bool(torch.equal(got_img, one_img) and torch.equal(got_rng, one_rng)))
            del got_img, got_rng, one_img, one_rng
        if world > 1:
            t = torch.tensor([ok, state["converted"]], dtype=torch.int64, device=cdev)
            lo = t.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            ok = int(lo[0].item())
        return bool(ok)

    def timed(zfar, steps, warmup):
        h.set_view(-180.0, 180.0, znear=ZNEAR, zfar=zfar)
        if multi:
            S["ex"] = None      # another scene: the strips' common capacity is agreed on again (outside the clock)
            drop_series()
        kern = []
        if c_loop:
            c_run(max(warmup, 1), check=True)
            S["rs"].sync()
            fence()
            t0 = time.perf_counter()
            if steps > 1:
                c_run(steps - 1)        # one call: returns when the panoramas are queued
            host_us["c_loop_call_us_per_panorama"] = (time.perf_counter() - t0) * 1e6 / max(steps - 1, 1)
            h.set_profiling(os.environ.get("BENCH_NO_KERNEL_EVENTS") is None)
            c_run(1)
            S["rs"].sync()      # every one of the K panoramas is assembled on its gathering rank ...
            fence()             # ... before the clock stops
            dt = time.perf_counter() - t0
            h.set_profiling(False)
        else:
            for _ in range(warmup):
                step()
            drain()
            fence()
            for key in host_us:
                host_us[key] = 0.0
            t0 = time.perf_counter()
            for k in range(steps):
                # HIP events around the kernels of the LAST of the K panoramas only: every event is a packet the command
                # processor works through between two kernels (profiling all K renders costs 1 % of the throughput)
                h.set_profiling(k == steps - 1 and os.environ.get("BENCH_NO_KERNEL_EVENTS") is None)
                step()              # N > 1: nothing in here waits on the host for the device (sharding.StripExchange)
            drain()                 # every one of the K panoramas is assembled on rank 0 ...
            h.sync()                # ... and, N = 1, converted ...
            fence()                 # ... before the clock stops
            dt = time.perf_counter() - t0
        if (not multi or S["SW"] > 0) and h.last_times() is not None:
            kern.append(h.last_times())         # HIP events of the last of the K panoramas
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kern

    if world > 1:
        h.set_view(-180.0, 180.0, znear=ZNEAR, zfar=args.zfar)
        rebalance()
    # The headline is the COLD draw: every vertex of every panorama transformed in full, as for a viewer that moves with
    # every render.  The library's vertex cache (hz_options_t::vertex_cache: from the second draw from a viewpoint on, the
    # view-independent half of the transform is read back from HBM) would serve K renders of one view from the cache: it is
    # switched off for `value` and measured beside it ("same_viewpoint").
    h.set_options(vertex_cache=0)
    dt, kern = timed(args.zfar, args.steps, args.warmup)
    verified = verify()

    # "parity gates recorded with every timing".  The panorama that was just timed - the LAST of
    # the K queued back to back, i.e. drawn by the kernels of a series: second round with coarse depth (k_march<false,
    # true>, k_hiz, k_big's chunk drop), reads before the atomics - is kept and compared below: with the oracle's render
    # of the same workload (the cpu_baseline leg computes it anyway) and with the SHA-256 of what the reference's own
    # shaders drew for this scene on Mesa llvmpipe (tests/golden/render_checksums.json; the reference's draw:
    # horizonator-lib.c:887-899).  A false entry makes the exit code non-zero.
    import hashlib

    def golden(zfar):
        """the committed reference render of this workload, if there is one and the DEM is the one it was made on"""
        name = {("cfg3", 600000.0): "cfg3_7x7_16000x4000", ("cfg3", 40000.0): "cfg3_7x7_16000x4000_zfar40km",
                ("cfg2", 600000.0): "cfg2_3x3_8000x2000"}.get((args.config, float(zfar)))
        try:
            c = json.load(open(os.path.join(ROOT, "tests", "golden", "render_checksums.json")))[name]
        except Exception:
            return None
        if S.get("mosaic_sha") is None:
            S["mosaic_sha"] = hashlib.sha256(np.ascontiguousarray(h.mosaic()).tobytes()).hexdigest()
        if S["mosaic_sha"] != c["mosaic_sha256"] or (c["W"], c["H"], c["R"], c["znear"]) != (W, H, R, ZNEAR):
            return None
        return c

    def keep_last(zfar):
        """the last timed panorama (N = 1: what d_img / d_rng hold now) and the plan of its draw"""
        if world != 1 or args.exchange_anyway or args.raster == 1:
            return None
        plan = h.last_plan()
        img = d_img.cpu().numpy()
        g = golden(zfar)
        return {"img": img, "rng": d_rng.cpu().numpy(), "plan": plan,
                "bgr_sha_is_llvmpipe": (hashlib.sha256(img.tobytes()).hexdigest() == g["bgr_sha256"]) if g else None}

    last = keep_last(args.zfar)

    # What ONE panorama costs (N = 1): the series above is K panoramas queued back to back, whose rounds and conversions
    # overlap; a caller that waits for each render sees this instead.  Outputs left in HBM, like `value`.
    single_latency = None
    if world == 1 and not args.exchange_anyway:
        ts1 = []
        for k in range(8):
            drain()
            fence()
            t1 = time.perf_counter()
            step()
            drain()
            h.sync()
            fence()
            ts1.append(time.perf_counter() - t1)
        single_latency = {"value": float(np.median(ts1[2:])) * 1e3, "samples_ms": [round(x * 1e3, 4) for x in ts1],
                          "what": "one render waited for, nothing queued before or behind it, outputs left in HBM (cold draw: vertex "
                                  "cache off); median of 6 after 2 warm-ups"}

    same_viewpoint = None
    if world == 1 and not args.no_extra and not args.exchange_anyway and args.raster != 1:
        h.set_options(vertex_cache=1)
        dt_c, kern_c = timed(args.zfar, args.steps, max(args.warmup, 3))
        plan_c = h.last_plan()
        img_c = d_img.cpu().numpy()
        same_viewpoint = {"ms_per_step": dt_c / args.steps * 1e3, "value": W * H * args.steps / dt_c / 1e6, "unit": "Mpix/s",
                          "from_vertex_cache": bool(plan_c.get("vertex_cache")),
                          "kernel_ms": (kern_c[-1]["raster_ms"] if kern_c else None),
                          "equals_the_cold_render": bool(last is not None and np.array_equal(img_c, last["img"]) and np.array_equal(d_rng.cpu().numpy(), last["rng"])),
                          "what": "the same K renders with the library's default: the viewer has not moved, so from the second draw on the "
                                  "view-independent half of every vertex's transform (two atan, two square roots; 16 B per vertex, %.2f GB) "
                                  "is read from HBM instead of computed - what a caller that turns or zooms sees; `value` above is the cold draw" % (16.0 * N * N / 1e9)}
        del img_c
        h.set_options(vertex_cache=0)

    # N > 1: the line's `value` is the throughput of --gather's mode (default rotate: panorama k is assembled on rank
    # k mod N, the panoramas stay spread over the ranks).  Beside it: the other mode (root0: north_star's "gather of the
    # strips" to ONE place, every panorama assembled on rank 0), and what ONE panorama takes at this N from the call to
    # the assembled panorama on rank 0 - a draw that is waited for, its gather, its conversion, nothing to overlap with.
    multi_extra = {}
    if world > 1 or args.exchange_anyway:
        def relayout():
            cos_lat = float(np.cos(np.radians(LAT)))
            apply_layout(balanced_layout(azimuth_density(W, -180.0, 180.0, cos_lat, floor=0.1), world, S["weights"]))
            h.set_view(-180.0, 180.0, znear=ZNEAR, zfar=args.zfar)
            if world > 1:
                rebalance()
        other = "root0" if args.gather == "rotate" else "rotate"
        set_gather(other)
        relayout()
        n2 = max(4, args.steps // 2)
        dt2, _ = timed(args.zfar, n2, 2)
        multi_extra["gather_" + other] = {"value": W * H * n2 / dt2 / 1e6, "unit": "Mpix/s", "ms_per_step": dt2 / n2 * 1e3, "steps": n2,
                                          "sector_widths": [c1 - c0 for c0, c1 in S["layout"]],
                                          "gathered_panorama_equals_single_gpu_render": verify()}
        if other != "root0":
            set_gather("root0")
            relayout()
        ts = []
        for k in range(6):
            drain()
            fence()
            t0 = time.perf_counter()
            step()
            drain()
            fence()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ts.append(float(t.item()))
        multi_extra["single_panorama_latency_ms"] = {"value": float(np.median(ts[1:])) * 1e3, "samples_ms": [x * 1e3 for x in ts],
                                                     "what": "one panorama at this N, gathered to rank 0 and converted there, from the call to the "
                                                             "assembled BGR8 + float32 range in rank 0's HBM; nothing queued before or behind it "
                                                             "(max over ranks, median of 5 after one warm-up)"}
    ms_per_step = dt / args.steps * 1e3
    value = W * H * args.steps / dt / 1e6

    # dominant kernel, measured live with HIP events on the render stream; N > 1: of the slowest rank
    names = ("raster_ms", "big_ms", "resolve_ms", "clear_ms", "near_ms", "total_ms")
    mine = [float(np.mean([k[n] for k in kern])) if kern else 0.0 for n in names]
    if world > 1:
        tk = torch.tensor(mine, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        mine = [float(x) for x in tk.tolist()]
    raster_ms, big_ms, resolve_ms, clear_ms, near_ms, total_ms = mine
    # algorithmic bytes of one render: int16 DEM read once +
    # BGR8 and float32 range written once; a sector accounts for its share
    algo_bytes = 2 * N * N + 7 * S["SW_max"] * H
    achieved = algo_bytes / (raster_ms * 1e-3) / 1e9 if raster_ms > 0 else 0.0
    traffic = None
    traffic_by_kernel = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get("config") == args.config and rec.get("zfar") == args.zfar and world == 1:
                # the whole render's HBM bytes (every kernel of one panorama of a series), and who moved them
                traffic = rec.get("hbm_bytes_per_render_all_kernels")
                traffic_by_kernel = {k: v.get("hbm_bytes_per_render") for k, v in rec.get("kernels", {}).items() if v.get("hbm_bytes_per_render")}
        except Exception:
            traffic = None
    traffic_note = ("RECORDED, not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command (FETCH_SIZE doubled: the "
                    "gfx950 correction, checked on the conversion's known reads), profiles/pmc_latest.json; `traffic` = HBM bytes of ALL kernels of one "
                    "render of a series, `traffic_by_kernel` says whose they are (the dominant kernel's own: k_march_far_round)") if traffic is not None else None

    # what actually bounds the dominant kernel: the SIMDs' vector issue slots.
    # Recorded counters of the same workload (profiles/), set against the duration measured now.
    valu = None
    mix = next((m for m in (os.path.join(ROOT, "profiles", "pmc_r%d_instruction_mix_cfg3.json" % r) for r in (6, 5, 4, 3, 2)) if os.path.exists(m)), "")
    if args.config == "cfg3" and args.zfar == 600000.0 and world == 1 and args.raster in (0, 2) and os.path.exists(mix):
        try:
            rec = json.load(open(mix))
            # (the second round's marching kernel; in a series of renders that is the instance with coarse depth, where the file has it)
            far = max((v for k, v in rec.items() if k.startswith("k_march_coarse_depth grid")), key=lambda v: v["SQ_INSTS_VALU"], default=None) \
                  or max((v for k, v in rec.items() if k.startswith("k_march grid")), key=lambda v: v["SQ_INSTS_VALU"])
            busy_ms = far["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * 2.4e9) * 1e3     # quad-cycles -> cycles, 1024 SIMDs at 2.4 GHz
            valu = {"recorded": True, "wave_instructions": far["SQ_INSTS_VALU"], "active_quad_cycles": far["SQ_ACTIVE_INST_VALU"],
                    "cycles_per_instruction": 4.0 * far["SQ_ACTIVE_INST_VALU"] / far["SQ_INSTS_VALU"],
                    "issue_busy_ms": busy_ms, "frac_of_kernel_ms": busy_ms / raster_ms,
                    "source": "counters RECORDED in profiles/" + os.path.basename(mix) + " (a rocprofv3 --pmc run of this workload), set against the "
                              "kernel duration measured live in this run; issue rates per instruction kind: profiles/valu_issue.json"}
            # ... and the roof that binds the RENDER, not one kernel: the vector-issue cycles of every kernel of one panorama of a
            # series (recorded, the same file) against the period measured now - all 1024 SIMDs busy for that share of the time
            per_render = {"k_march_coarse_depth grid": 1, "k_march grid 2": 1, "k_big grid": 2, "k_clip": 2, "k_hiz grid": 1, "k_resolve4": 1}
            quad = 0.0
            for key, v in rec.items():
                for prefix, launches in per_render.items():
                    if key.startswith(prefix) and isinstance(v, dict) and "SQ_ACTIVE_INST_VALU" in v:
                        quad += launches * v["SQ_ACTIVE_INST_VALU"]
            valu["whole_render"] = {"active_quad_cycles": quad, "issue_busy_ms": quad * 4.0 / (1024 * 2.4e9) * 1e3,
                                    "frac_of_ms_per_step": quad * 4.0 / (1024 * 2.4e9) * 1e3 / (dt / args.steps * 1e3),
                                    "what": "sum over the kernels of one panorama of a series (second-round k_march, the first round's, "
                                            "2 x k_big, 2 x k_clip, k_hiz, k_resolve4) of the recorded SQ_ACTIVE_INST_VALU, as time on "
                                            "1024 SIMDs at 2.4 GHz, over ms_per_step: the share of the chip's vector issue the render uses"}
        except Exception:
            valu = None

    extra = {}
    if not args.no_extra and abs(args.zfar - 40000.0) > 1:
        dt40, k40 = timed(40000.0, max(3, args.steps // 2), 1)
        n40 = max(3, args.steps // 2)
        last40 = keep_last(40000.0)
        extra["zfar_40km"] = {"value": W * H * n40 / dt40 / 1e6, "unit": "Mpix/s",
                              "ms_per_step": dt40 / n40 * 1e3,
                              "note": "API default far clip (reference horizonator.h:10); >95% of the mosaic is beyond it"}
        if last40 is not None:
            extra["zfar_40km"]["parity"] = {"bgr_sha_is_llvmpipe": last40["bgr_sha_is_llvmpipe"], "rounds": last40["plan"]["rounds"],
                                            "coarse_depth_in_series": last40["plan"]["coarse_depth"],
                                            "what": "the last of the timed panoramas: SHA-256 of its BGR bytes against the reference's shaders on llvmpipe"}
        del last40

    # the reference hands its results over in HOST memory.  The same panorama through
    # horizonator_render_offscreen() into caller-owned (pageable) buffers that the caller keeps, as
    # standalone.c does: device time + PCIe + the copy into the caller's pages.  Reported beside
    # `value`, never as it.
    host_incl = None
    if rank == 0 and world == 1 and not args.no_host:
        h.set_view(-180.0, 180.0, znear=ZNEAR, zfar=args.zfar)
        himg = np.zeros((H, W, 3), np.uint8)
        hrng = np.zeros((H, W), np.float32)
        h.render_device(d_img.data_ptr(), d_rng.data_ptr())
        h.sync()
        want_img, want_rng = d_img.cpu().numpy(), d_rng.cpu().numpy()
        # median of >= 10 calls after 2 warm-ups
        ts = []
        for k in range(12):
            t0 = time.perf_counter()
            h.render_into(himg, hrng)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts[2:]))
        same = bool(np.array_equal(himg, want_img) and np.array_equal(hrng, want_rng))
        # a viewer that MOVES between calls, with the library's defaults (vertex cache on): every call is a first draw from
        # its viewpoint - no call may pay for a cache fill (the sectors of one call are one draw)
        h.set_options(vertex_cache=1)
        ts_move = []
        for k in range(8):
            h.set_view(-180.0, 180.0, lat=LAT + 1e-4 * (k + 1), lon=LON, znear=ZNEAR, zfar=args.zfar)
            t0 = time.perf_counter()
            h.render_into(himg, hrng)
            ts_move.append(time.perf_counter() - t0)
        moved_fill = bool(h.last_plan().get("vertex_cache"))
        h.set_options(vertex_cache=0)
        h.set_view(-180.0, 180.0, lat=LAT, lon=LON, znear=ZNEAR, zfar=args.zfar)
        ts_fresh = []
        for k in range(8):          # what the reference's Python wrapper does: new arrays (untouched pages) for every call
            t0 = time.perf_counter()
            fresh = (np.empty((H, W, 3), np.uint8), np.empty((H, W), np.float32))
            h.render_into(*fresh)
            ts_fresh.append(time.perf_counter() - t0)
            if k == 7:
                same = same and bool(np.array_equal(fresh[0], want_img) and np.array_equal(fresh[1], want_rng))
            del fresh
        t_fresh = float(np.median(ts_fresh[2:]))
        ts_py = []
        for k in range(8):          # the Python mirror's render(): its arrays are made of the memory of results the caller dropped
            t0 = time.perf_counter()
            res = h.render(-180.0, 180.0, znear=ZNEAR, zfar=args.zfar)
            ts_py.append(time.perf_counter() - t0)
            if k == 7:
                same = same and bool(np.array_equal(res[0], want_img) and np.array_equal(res[1], want_rng))
            del res
        t_py = float(np.median(ts_py[2:]))
        # a caller that renders a series with two sets of buffers: begin k+1, then end k (include/horizonator_amd.h)
        himg2 = np.zeros((H, W, 3), np.uint8)
        hrng2 = np.zeros((H, W), np.float32)
        bufs = ((himg, hrng), (himg2, hrng2))
        himg[:] = 0; hrng[:] = 0
        nser = 14
        h.render_begin(*bufs[0])
        marks = []
        for k in range(1, nser + 1):
            if k < nser:
                h.render_begin(*bufs[k % 2])
            h.render_end()
            marks.append(time.perf_counter())
        t_series = float(np.median(np.diff(marks[2:])))
        same_series = bool(np.array_equal(himg, want_img) and np.array_equal(hrng, want_rng) and
                           np.array_equal(himg2, want_img) and np.array_equal(hrng2, want_rng))
        opts = h.options()
        host_incl = {"ms": t * 1e3, "value": W * H / t / 1e6, "unit": "Mpix/s", "results_GBps": 7 * W * H / t / 1e9,
                     "what": "horizonator_render_offscreen() into the caller's pageable host buffers (BGR8 + float32 range), one call "
                             "waited for - SURVEY.md 8(d)'s metric as the reference's callers see it: the median of 10 calls after 2 warm-ups.  "
                             "The panorama is drawn and shipped in azimuth sectors (sector s+1 drawn while sector s crosses PCIe); the terrain "
                             "pixels only travel, 4 B each (the sky, 62 % of this image, is constants the host threads fill in while the "
                             "device draws; ranges are made of the depths on the host)",
                     "ms_all_calls": [round(x * 1e3, 3) for x in ts],
                     "first_call_ms": ts[0] * 1e3,
                     "first_call": "the first horizonator_render_offscreen() of this context into host memory (buffers from np.zeros: "
                                   "untouched pages; the pool, the transfer stream and the pinned landing were made by horizonator_init)",
                     "moving_viewer_ms": float(np.median(ts_move[2:])) * 1e3, "moving_viewer_used_vertex_cache": moved_fill,
                     "moving_viewer": "the same call with the library's default options from a viewpoint that moves by 1e-4 degrees of "
                                      "latitude between calls (median of 6 after 2)",
                     "ms_with_fresh_arrays_per_call": t_fresh * 1e3,
                     "ms_python_render": t_py * 1e3,
                     "python_render": "horizonator_amd.horizonator.render() in a loop that drops its results (the mirror of the reference's "
                                      "Python API): the arrays it returns are made of the memory of results the caller let go of - pages mapped "
                                      "already - where ms_with_fresh_arrays_per_call hands np.empty() arrays to every call, as the reference's "
                                      "wrapper does (horizonator-pywrap.c:234-250)",
                     "ms_per_panorama_two_in_flight": t_series * 1e3,
                     "two_in_flight": "horizonator_amd_render_begin / _end with two sets of buffers: begin k+1, then end k - the device draws one "
                                      "panorama while the other crosses PCIe; median interval between ends over a series of %d" % nser,
                     "host_sectors": opts.get("host_sectors", 0) or ("auto: 4" if W * H >= 32e6 else "auto: 2" if W * H >= 12e6 else "auto: 1"),
                     "copy_threads": int(os.environ.get("HZ_COPY_THREADS", 0)) or (min(32, (os.cpu_count() or 8) // 8) if (os.cpu_count() or 8) >= 32 else 4),
                     "equals_device_render": same, "two_in_flight_equals_device_render": same_series}
        del himg2, hrng2, want_img, want_rng
        del himg, hrng

    # horizonator_init() as a number of its own: the first context of this process (above: the HIP runtime's
    # own start-up, the first allocations and - in this harness - tiles written a moment ago are in it) and one more of the same
    # configuration made now
    init_again_s = None
    init_again_all = []
    if rank == 0 and world == 1 and not args.no_extra:
        for _ in range(2):          # (twice: on the test boxes the first open() of the tiles after minutes of other work takes 2 ms a file, not ours)
            t0 = time.perf_counter()
            h2 = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R, SRTM1=cfg.get("srtm1", False))
            init_again_all.append(time.perf_counter() - t0)
            h2.close()
            del h2
        init_again_s = min(init_again_all)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the CPU restatement (oracle/, test infrastructure) timed on this box's
        # host cores on the same workload, as the checker's throughput - one render
        import oracle
        od = oracle.Dem(LAT, LON, dems, radius_cells=R)
        mosaic = od.mosaic()
        v = od.view(LAT, LON, W, H, -180.0, 180.0, znear=ZNEAR, zfar=args.zfar)
        cores = os.cpu_count() or 1
        samples = []
        for k in range(3):
            t0 = time.perf_counter()
            ref = oracle.render(mosaic, v, W, H, want=("bgr", "ranges"))
            samples.append(time.perf_counter() - t0)
            if k == 0 and last is not None:         # the checker's picture against the panorama that was timed (outside the clock)
                last["bgr_is_oracle"] = bool(np.array_equal(ref["bgr"], last["img"]))
                last["ranges_is_oracle"] = bool(np.array_equal(ref["ranges"], last["rng"]))
            del ref
        cdt = float(np.median(samples))
        cpu = {"value": W * H / cdt / 1e6, "unit": "Mpix/s", "cores": cores, "kind": "port",
               "sample": f"3 full {W}x{H} renders of the same workload by oracle/ (C + OpenMP, {cores} threads): "
                         + ", ".join(f"{x:.2f}" for x in samples) + " s; value = the median"}

    # the reference's own shaders on Mesa llvmpipe cannot run on the GPU box (neither /root/reference nor Mesa's
    # software driver is there): what tools/llvmpipe_timing.py measured in the build container, with its machine
    ref_rec = None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "llvmpipe_timing.json")))
        rows = [r for r in rec["rows"] if r["config"] == args.config and abs(r["zfar"] - args.zfar) < 1]
        if rows:
            ref_rec = {"value": rows[0]["mpix_per_s_draw_plus_readback"], "unit": "Mpix/s", "draw_s": rows[0]["draw_s"],
                       "cores": rec["machine"]["cores"], "cpu": rec["machine"]["cpu"],
                       "what": "reference vertex/geometry/fragment.glsl on Mesa llvmpipe, glClear+glDrawElements+2 glReadPixels per frame; "
                               "recorded in the build container by tools/llvmpipe_timing.py (profiles/llvmpipe_timing.json), not measured in this run"}
    except Exception:
        ref_rec = None

    # scenes the kernel was not tuned on: a few renders each, ms per render and picoseconds per
    # triangle of the mosaic; profiles/r3_scenes.json holds the same scenes with the waves' counters
    scene_recs = None
    if rank == 0 and world == 1 and not args.no_scenes and not args.no_extra:
        import scenes
        scene_recs = {}
        cache = {"key": (R, W, H, False, False), "h": h} if args.config == "cfg3" else {}
        for name in [n for n in scenes.DEFAULT if n != args.config]:
            try:
                scene_recs[name] = scenes.run_scene(name, steps=8, counters=False, cache=cache)
            except Exception as e:      # never lose the line over an extra
                scene_recs[name] = {"error": repr(e)}
        if cache.get("h") is not None and cache["h"] is not h:
            cache["h"].close()
        head = value and (1e12 / (value * 1e6) * W * H / (2 * (N - 1) ** 2))      # ps per triangle of the headline run
        for rec in scene_recs.values():
            if "ps_per_triangle" in rec and head:
                rec["ps_per_triangle_vs_headline"] = rec["ps_per_triangle"] / head

    if rank == 0:
        line = {
            "metric": "panorama Mpix/s (360deg, 16k-wide, 7x7 SRTM3 tiles)" if args.config == "cfg3"
                      else f"panorama Mpix/s (360deg, {W}x{H}, {cfg['tiles']})",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {cfg['tiles']}, R={R} ({N}x{N} samples, {2*(N-1)**2/1e6:.1f} M triangles), "
                            f"{W}x{H} 360deg panorama, znear {ZNEAR:g} m, zfar {args.zfar:g} m",
                "sector_widths": [c1 - c0 for c0, c1 in S["layout"]],
                "wire_bytes_per_rank": (4 * state["wire_words"] if sparse else 4 * H * S["SW_max"]) if multi else 0,
                "strip_resends": (S["ex"].resends if multi and sparse and S.get("ex") is not None else 0),
                "parallelism": f"azimuth sectors x{world}" + (" + " + ("RCCL" if args.backend == "nccl" else "gloo (diagnostic, through host memory)") + " gather of " + ("sparse (terrain pixels only + mask)" if sparse else "packed") + " depth+shade strips (4 B/pixel) to " + ("rank k mod N for panorama k" if args.gather == "rotate" else "rank 0") + ", overlapped with the next render; the gathering rank converts them to BGR8 + float32 range" if world > 1 else ""),
                "gather": args.gather if multi else None,
                "raster": {0: "auto", 1: "scatter", 2: "march"}.get(args.raster, f"experiment {args.raster}"),
                "outputs": "BGR8 + float32 range, device-resident",
                "init_s": init_s,
                "init": {"first_context_of_the_process_s": init_s, "another_context_s": init_again_s, "another_context_s_both": init_again_all,
                         "ingest": os.environ.get("HORIZONATOR_INGEST", "device"),
                         "what": "horizonator_init(): tiles mapped, device state, the tiles' bytes through pinned memory and k_ingest, the host "
                                 "path's threads and pinned landing area, one throw-away draw; HZ_INIT_TIMES=1 prints the parts (tools/init_times.py)"},
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_by_kernel": traffic_by_kernel, "traffic_source": traffic_note,
                "kernel": "k_scatter" if args.raster == 1 else "k_march", "kernel_ms": raster_ms,
                "kernel_launch": ("second round of a two-round draw: every strip but those next to the viewer (the first round's "
                                  "k_march is in other_kernels_ms.round1_near_viewer with its queue kernels)") if near_ms > 0.05 else "the draw's only k_march launch",
                "frac_whole_render": algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes": algo_bytes,
                "other_kernels_ms": {"clear": clear_ms, "round1_near_viewer": near_ms, "queues_after": big_ms, "resolve": resolve_ms},
                "device_ms_per_render_sum_of_stages": total_ms,
                "achieved_whole_render": algo_bytes / (ms_per_step * 1e-3) / 1e9,
                "valu_issue": valu,
            },
            "cpu_baseline": cpu,
            "reference_llvmpipe_recorded": ref_rec,
        }
        if single_latency is not None:
            line["single_panorama_latency_ms"] = single_latency
        if same_viewpoint is not None:
            line["same_viewpoint"] = same_viewpoint
        if host_incl is not None:
            line["host_inclusive"] = host_incl
        if scene_recs is not None:
            line["scenes"] = scene_recs
        if verified is not None:
            line["gathered_panorama_equals_single_gpu_render"] = verified
        line.update(multi_extra)
        if last is not None:
            line["parity"] = {"bgr": last.get("bgr_is_oracle"), "ranges": last.get("ranges_is_oracle"),
                              "bgr_sha_is_llvmpipe": last["bgr_sha_is_llvmpipe"],
                              "coarse_depth_in_series": last["plan"]["coarse_depth"], "rounds": last["plan"]["rounds"],
                              "what": "the LAST of the K timed panoramas (a draw of a series: the kernel instances the clock saw), every BGR byte and "
                                      "every float32 range against oracle/'s render of the same workload (null: --no-cpu-baseline), and the SHA-256 of "
                                      "its BGR bytes against what the reference's vertex/geometry/fragment.glsl drew on Mesa llvmpipe "
                                      "(tests/golden/render_checksums.json; null: no reference render of this workload is committed)"}
        if c_loop:
            line["loop"] = {"driver": "horizonator_rccl_render_series (C, include/horizonator_rccl.h)",
                            "host_us_per_panorama": host_us.get("c_loop_call_us_per_panorama")}
        if os.environ.get("BENCH_HOST_TIMES") and host_us["n"]:
            line["host_us_per_step"] = {k: v / host_us["n"] for k, v in host_us.items() if k != "n"}
        line.update(extra)
        print(json.dumps(line), flush=True)
        gates = [line.get("parity", {}).get(k) for k in ("bgr", "ranges", "bgr_sha_is_llvmpipe")]
        gates.append(extra.get("zfar_40km", {}).get("parity", {}).get("bgr_sha_is_llvmpipe"))
        gates.append(line.get("gathered_panorama_equals_single_gpu_render"))
        gates += [v.get("gathered_panorama_equals_single_gpu_render") for k, v in multi_extra.items() if k.startswith("gather_")]
        gates.append(host_incl["equals_device_render"] if host_incl is not None else None)
        gates.append(same_viewpoint["equals_the_cold_render"] if same_viewpoint is not None and last is not None else None)
        gates.append(host_incl["two_in_flight_equals_device_render"] if host_incl is not None else None)
        failed = any(g is False for g in gates)
    drop_series()
    h.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0 and failed:
        sys.exit("bench.py: a parity gate of the line above is false - the timing is void")


if __name__ == "__main__":
    main()
