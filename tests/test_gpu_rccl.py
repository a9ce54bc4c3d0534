"""include/horizonator_rccl.h: the C-level exchange over RCCL.  One GPU is what a test box has, so the
communicator has one rank - RCCL runs its send/receive path all the same (a rank sending to itself inside
a group), and the strip that went through it must convert to the panorama a plain render gives."""
import ctypes as C
import os

import numpy as np
import pytest

import hzutil

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_strips_through_rccl_convert_to_the_same_panorama():
    import torch
    import horizonator_amd
    from horizonator_amd import _lib
    from horizonator_amd.sharding import sparse_header_words, sparse_mask_stride
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
    _lib.load()
    hz = C.CDLL(os.path.join(ROOT, "horizonator_amd", "libhorizonator_rccl.so"))
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    hz.horizonator_rccl_gather_strips.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                  C.c_size_t, C.POINTER(C.c_void_p), C.c_void_p]
    hz.horizonator_rccl_broadcast_mosaic.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    torch.cuda.set_device(0)
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0

    LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
    R, W, H = 300, 1000, 250
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    try:
        image, ranges = h.render(-180, 180, zfar=30000.0)
        ms = sparse_mask_stride(W)
        hdr = sparse_header_words(H, ms)
        dev = torch.device("cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        # (torch.empty: the library zeroes the strip's count word itself, on its own stream; a torch.zeros would
        # fill the buffer on torch's stream, unordered against the render that writes it)
        d_send = torch.empty(hdr + H * W, dtype=torch.int32, device=dev)
        d_recv = torch.full((hdr + H * W,), -1, dtype=torch.int32, device=dev)
        h.render_sparse(d_send.data_ptr(), ms)
        bins = (C.c_void_p * 1)(d_recv.data_ptr())
        rc = hz.horizonator_rccl_gather_strips(C.byref(h._ctx), comm, 0, 1, 0, d_send.data_ptr(), d_send.numel(), bins, stream)
        assert rc == 0
        h.waits_for_stream(stream)                              # the conversion runs behind the arrival, on the device
        d_img = torch.full((H, W, 3), 77, dtype=torch.uint8, device=dev)
        d_rng = torch.full((H, W), -7.0, dtype=torch.float32, device=dev)
        h.resolve_sparse_gathered([(d_recv.data_ptr(), 0, W)], ms, d_img.data_ptr(), d_rng.data_ptr())
        h.sync()
        torch.cuda.synchronize()
        assert int(d_recv[0]) == int((ranges > 0).sum())
        assert np.array_equal(d_img.cpu().numpy(), image)
        assert np.array_equal(d_rng.cpu().numpy(), ranges)
        # the mosaic broadcast (a no-op of one rank, through RCCL all the same)
        m = torch.from_numpy(h.mosaic().view(np.uint8).copy()).to(dev)
        before = m.clone()
        assert hz.horizonator_rccl_broadcast_mosaic(comm, 0, m.data_ptr(), 2 * R, stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(m, before)
    finally:
        h.close()
        rccl.ncclCommDestroy(comm)


def _series_case(sector=None):
    import torch
    import horizonator_amd
    from horizonator_amd.sharding import RcclSeries
    LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
    R, W, H = 300, 1000, 250
    dev = torch.device("cuda:0")
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    rs = None
    try:
        image, ranges = h.render(-180, 180, zfar=30000.0)
        c0, c1 = sector if sector else (0, W)
        h.set_sector(c0, c1)
        d_img = torch.full((H, W, 3), 77, dtype=torch.uint8, device=dev)
        d_rng = torch.full((H, W), -7.0, dtype=torch.float32, device=dev)
        terrain = int((ranges[:, c0:c1] > 0).sum())
        yield h, image, ranges, d_img, d_rng, terrain, (c0, c1), dev
    finally:
        h.close()


def test_a_series_of_panoramas_driven_from_c_gives_the_same_panorama():
    """horizonator_rccl_render_series (include/horizonator_rccl.h): draw, gather through RCCL and conversion of the gathered
    strip for several panoramas in one C call, two strip buffers in turn - the panorama left in the outputs is the plain render"""
    import torch
    from horizonator_amd.sharding import RcclSeries
    for rotate in (False, True):
        for h, image, ranges, d_img, d_rng, terrain, (c0, c1), dev in _series_case():
            H, W = ranges.shape
            rs = RcclSeries(h, [(0, W)], H, 10**9, d_img.data_ptr(), d_rng.data_ptr(), rotate, dev, nslots=2)
            try:
                assert rs.run(3)                        # (returns at once: nothing waits on the host)
                assert rs.run(2, check_fit=True)
                rs.sync()
                assert [int(t[0]) for t in rs.strips] == [terrain, terrain]
                assert np.array_equal(d_img.cpu().numpy(), image)
                assert np.array_equal(d_rng.cpu().numpy(), ranges)
            finally:
                rs.close()


def test_a_series_reports_a_strip_that_did_not_fit_and_a_sector_lands_in_its_columns():
    import torch
    from horizonator_amd.sharding import RcclSeries, sparse_header_words, sparse_mask_stride
    for h, image, ranges, d_img, d_rng, terrain, (c0, c1), dev in _series_case(sector=(300, 700)):
        H, W = ranges.shape
        hdr = sparse_header_words(H, sparse_mask_stride(c1 - c0))
        assert terrain > 2000
        rs = RcclSeries(h, [(c0, c1)], H, hdr + terrain - 1000, d_img.data_ptr(), d_rng.data_ptr(), False, dev, nslots=2)
        try:
            assert rs.run(3, check_fit=True) is False           # 1000 words short: reported, nothing hangs
        finally:
            rs.close()
        d_img.fill_(77); d_rng.fill_(-7.0)
        rs = RcclSeries(h, [(c0, c1)], H, hdr + terrain + 16, d_img.data_ptr(), d_rng.data_ptr(), False, dev, nslots=1)
        try:
            assert rs.run(4, check_fit=True) is True
            rs.sync()
            got_img, got_rng = d_img.cpu().numpy(), d_rng.cpu().numpy()
            assert np.array_equal(got_img[:, c0:c1], image[:, c0:c1]) and np.array_equal(got_rng[:, c0:c1], ranges[:, c0:c1])
            assert (got_img[:, :c0] == 77).all() and (got_rng[:, c1:] == -7.0).all()     # (nothing outside the strip's columns is touched)
        finally:
            rs.close()
