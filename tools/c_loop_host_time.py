"""what the host spends per panorama in the N > 1 loop, on the one GPU a test box has: a 1/8 (and a 1/4) sector of cfg3 drawn as a
sparse strip, sent through RCCL (a communicator of one rank: the send/receive path runs all the same) and converted into the
full-width outputs - driven by horizonator_rccl_render_series (C, include/horizonator_rccl.h) and by the same calls made
from Python one by one (what bench.py's loop does, without torch.distributed's part: that is in bench.py's BENCH_HOST_TIMES)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hzutil, horizonator_amd
from horizonator_amd.sharding import RcclSeries, sector_columns
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 4200, 16000, 4000
dev = torch.device("cuda:0")
h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
h.set_view(-180, 180, zfar=600000.0)
d_img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev); d_rng = torch.empty((H, W), dtype=torch.float32, device=dev)
for G in (8, 4, 1):
    c0, c1 = sector_columns(W, G, 1 if G > 1 else 0)
    h.set_sector(c0, c1)
    from horizonator_amd.sharding import sparse_header_words, sparse_mask_stride
    ms_ = sparse_mask_stride(c1 - c0); hdr = sparse_header_words(H, ms_)
    probe = torch.empty(hdr + H * (c1 - c0), dtype=torch.int32, device=dev)
    h.render_sparse(probe.data_ptr(), ms_); h.sync()
    words = hdr + int(1.1 * int(probe[0])) + 1024             # what the ranks would agree on (sharding.agree_on_capacity)
    del probe
    rs = RcclSeries(h, [(c0, c1)], H, words, d_img.data_ptr(), d_rng.data_ptr(), False, dev, nslots=2)
    rs.run(8); rs.sync()
    n = 200
    t0 = time.perf_counter(); rs.run(n); t1 = time.perf_counter(); rs.sync(); t2 = time.perf_counter()
    c_host, c_dev = (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
    # the same steps, one library call at a time from Python
    st = rs.stream.cuda_stream
    hz = rs._hz
    import ctypes as C
    hz.horizonator_rccl_gather_strips.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p]
    def one(i):
        slot = i % 2
        h.waits_for_stream(st)
        h.render_sparse(rs.strips[slot].data_ptr(), rs.mask_stride)
        bins = (C.c_void_p * 1)(rs.bins[slot].data_ptr())
        assert hz.horizonator_rccl_gather_strips(C.byref(h._ctx), rs.comm, 0, 1, 0, rs.strips[slot].data_ptr(), rs.words, bins, st) == 0
        h.waits_for_stream(st)
        h.resolve_sparse_gathered([(rs.bins[slot].data_ptr(), c0, c1 - c0)], rs.mask_stride, d_img.data_ptr(), d_rng.data_ptr())
    for i in range(8): one(i)
    rs.sync()
    t0 = time.perf_counter()
    for i in range(n): one(i)
    t1 = time.perf_counter(); rs.sync(); t2 = time.perf_counter()
    print(f"1/{G} of cfg3 ({4 * rs.words / 1e6:.1f} MB per strip through RCCL): C loop {c_host:.0f} us of host time per panorama (device: a panorama every {c_dev:.0f} us); "
          f"the same calls from Python {(t1 - t0) / n * 1e6:.0f} us (device {(t2 - t0) / n * 1e6:.0f} us)", flush=True)
    rs.close()
h.close()
