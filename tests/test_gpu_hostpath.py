"""horizonator_render_offscreen() as the reference's callers see it - results in HOST memory (reference
horizonator-lib.c:911-1051) - through hz_hostpath.cpp: the panorama drawn and shipped in azimuth sectors, 4 bytes per terrain
pixel over PCIe, the readback conversion (reference :1006-1047) on the host's vector unit, the sky filled in by host threads.
Whatever the number of sectors, the outputs asked for and the order of begin / end, the caller's buffers hold the bytes the
oracle computes."""
import ctypes as C

import numpy as np
import pytest

import hzutil
import oracle

pytestmark = pytest.mark.gpu
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON


@pytest.fixture(scope="module")
def scene():
    import horizonator_amd
    R, W, H = 500, 3001, 750                                    # an odd width: sector edges and blobs that do not end on a multiple of 4
    d = hzutil.dem_dir_for(LAT, LON, R)
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    yield h, od, W, H
    h.close()


def _want(od, W, H, az0, az1, zfar, lat=LAT, lon=LON):
    return oracle.render(od.mosaic(), od.view(lat, lon, W, H, az0, az1, zfar=zfar), W, H)


@pytest.mark.parametrize("sectors", [1, 2, 3, 4, 8])
def test_any_number_of_sectors_delivers_the_oracles_bytes(scene, sectors):
    h, od, W, H = scene
    h.set_options(host_sectors=sectors)
    try:
        for az0, az1, zfar in ((-180.0, 180.0, 200000.0), (20.0, 140.0, 40000.0)):
            want = _want(od, W, H, az0, az1, zfar)
            image, ranges, index, z24 = h.render_full(az0, az1, zfar=zfar)
            hzutil.assert_same_render(dict(bgr=image, ranges=ranges, index=index, z24=z24), want, f"{sectors} sectors, az [{az0},{az1}]")
            # the reference's own call: image + ranges; and each alone (other blob formats: the shade only, no index)
            image2, ranges2 = h.render(az0, az1, zfar=zfar)
            assert np.array_equal(image2, want["bgr"]) and np.array_equal(ranges2, want["ranges"])
            assert np.array_equal(h.render(az0, az1, zfar=zfar, return_range=False), want["bgr"])
            assert np.array_equal(h.render(az0, az1, zfar=zfar, return_image=False), want["ranges"])
        # readers of the framebuffer after a sectored call see the whole view (the last sector's framebuffer is not it)
        ys, xs = np.nonzero(index >= 0)
        for k in (0, len(ys) // 3, len(ys) - 1):
            assert h.pick(int(xs[k]), int(ys[k])) is not None
        sky = np.argwhere(index < 0)[0]
        assert h.pick(int(sky[1]), int(sky[0])) is None
    finally:
        h.set_options(host_sectors=0)


@pytest.mark.parametrize("prefill", [0, 37, 100])
@pytest.mark.parametrize("sectors", [1, 3])
def test_any_share_of_sky_filled_beforehand(scene, sectors, prefill, monkeypatch):
    """HZ_HOST_PREFILL: the rows that get their sky before the blobs arrive (the others: a blob writes the sky pixels of its own
    tile, tiles without a blob are filled when their sector has been walked) - every byte of every buffer is written either
    way: the buffers start out as rubbish"""
    h, od, W, H = scene
    monkeypatch.setenv("HZ_HOST_PREFILL", str(prefill))
    h.set_options(host_sectors=sectors)
    try:
        for az0, az1, zfar in ((-180.0, 180.0, 200000.0), (-100.0, -10.0, 3000.0)):      # (the second: a close far clip - sectors and tiles without any terrain)
            want = _want(od, W, H, az0, az1, zfar)
            h.set_view(az0, az1, zfar=zfar)
            image = np.full((H, W, 3), 0x5A, np.uint8); ranges = np.full((H, W), 123.0, np.float32)
            h.render_into(image, ranges)
            assert np.array_equal(image, want["bgr"]) and np.array_equal(ranges, want["ranges"]), (sectors, prefill, az0)
            image[:] = 0x5A
            h.render_into(image, None)
            assert np.array_equal(image, want["bgr"])
    finally:
        h.set_options(host_sectors=0)
        h.set_view(-180.0, 180.0, zfar=40000.0)


def test_two_panoramas_in_flight(scene):
    """begin k+1 before end k: different views, buffers of their own, ended in the order begun"""
    h, od, W, H = scene
    if h.options()["host_dense"]:
        # (tools/gpu_modes.sh runs the suite with HZ_HOST_DENSE=1 too: such a context delivers with the synchronous call only, and says so)
        with pytest.raises(RuntimeError):
            h.render_begin(np.empty((H, W, 3), np.uint8), np.empty((H, W), np.float32))
        return
    views = [(-180.0, 180.0, 200000.0, LAT, LON), (-100.0, 100.0, 60000.0, LAT + 0.02, LON - 0.03), (-180.0, 180.0, 9000.0, LAT - 0.01, LON)]
    for sectors in (0, 3):
        h.set_options(host_sectors=sectors)
        bufs = [(np.empty((H, W, 3), np.uint8), np.empty((H, W), np.float32)) for _ in views]
        try:
            pending = []
            for k, (az0, az1, zfar, lat, lon) in enumerate(views):
                h.set_view(az0, az1, lat=lat, lon=lon, zfar=zfar)
                h.render_begin(*bufs[k])
                pending.append(k)
                if len(pending) == 2:
                    h.render_end(); pending.pop(0)
            while pending:
                h.render_end(); pending.pop(0)
            with pytest.raises(RuntimeError):
                h.render_end()                                  # nothing is in flight
            for k, (az0, az1, zfar, lat, lon) in enumerate(views):
                want = _want(od, W, H, az0, az1, zfar, lat, lon)
                assert np.array_equal(bufs[k][0], want["bgr"]), (sectors, k)
                assert np.array_equal(bufs[k][1], want["ranges"]), (sectors, k)
            # a third begin while two are in flight is refused, and a synchronous render while one is
            h.render_begin(*bufs[0]); h.render_begin(*bufs[1])
            with pytest.raises(RuntimeError):
                h.render_begin(*bufs[2])
            with pytest.raises(RuntimeError):
                h.render(-180.0, 180.0)
            h.render_end(); h.render_end()
        finally:
            h.set_options(host_sectors=0)
    h.set_view(-180.0, 180.0, lat=LAT, lon=LON)


def test_sectors_of_a_sector_context_and_small_images(scene):
    """a context that is itself one sector of a panorama (multi-GPU) is not cut further; its buffers have the sector's width"""
    h, od, W, H = scene
    want = _want(od, W, H, -180.0, 180.0, 200000.0)
    h.set_options(host_sectors=4)
    h.set_sector(701, 1502)
    try:
        image, ranges, index, z24 = h.render_full(-180.0, 180.0, zfar=200000.0)
        assert image.shape == (H, 801, 3)
        for k, a in (("bgr", image), ("ranges", ranges), ("index", index), ("z24", z24)):
            assert np.array_equal(a, want[k][:, 701:1502]), k
    finally:
        h.set_sector(0, W)
        h.set_options(host_sectors=0)


def test_options_round_trip(scene):
    h, _, _, _ = scene
    o = h.options()
    assert set(o) >= {"host_sectors", "adapt", "fast_math", "rounds"}
    h.set_options(rounds=2, near_cells=40)
    assert h.options()["rounds"] == 2 and h.options()["near_cells"] == 40
    h.set_options(rounds=o["rounds"], near_cells=o["near_cells"])
    assert h.options() == o
    with pytest.raises(TypeError):
        h.set_options(no_such_thing=1)


def test_a_context_closed_with_a_panorama_in_flight_finishes_it_first():
    """render_begin() and then close() - the pool's tasks name the caller's buffers and the job's counters, the
    copies write its landing area: the context ends the panorama itself before it frees anything"""
    import horizonator_amd
    R, W, H = 300, 2048, 512
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    want = oracle.render(od.mosaic(), od.view(LAT, LON, W, H, -180.0, 180.0, zfar=100000.0), W, H)
    for n in (1, 2):
        h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
        if h.options()["host_dense"]:
            h.close()
            return
        h.set_view(-180.0, 180.0, zfar=100000.0)
        bufs = [(np.zeros((H, W, 3), np.uint8), np.zeros((H, W), np.float32)) for _ in range(n)]
        for b in bufs:
            h.render_begin(*b)
        h.close()                                               # no render_end()
        for b in bufs:
            assert np.array_equal(b[0], want["bgr"]) and np.array_equal(b[1], want["ranges"])


def test_the_sectors_of_a_call_are_one_draw_for_the_vertex_cache(scene):
    """a viewer that moves between calls never pays for a fill of the vertex cache (every call is the FIRST draw
    from its viewpoint, however many sectors it is drawn in); a viewer that stays gets it with the second call"""
    h, od, W, H = scene
    if not h.options()["vertex_cache"]:
        return
    h.set_options(host_sectors=4)
    try:
        img, rng = np.empty((H, W, 3), np.uint8), np.empty((H, W), np.float32)
        for k in range(4):
            h.set_view(-180.0, 180.0, lat=LAT + 2e-4 * (k + 1), lon=LON, zfar=60000.0)
            h.render_into(img, rng)
            assert not h.last_plan()["vertex_cache"], k
        want = _want(od, W, H, -180.0, 180.0, 60000.0, LAT + 8e-4, LON)
        assert np.array_equal(img, want["bgr"]) and np.array_equal(rng, want["ranges"])
        h.render_into(img, rng)                                 # the second call from there: filled, and read from it
        assert h.last_plan()["vertex_cache"]
        assert np.array_equal(img, want["bgr"]) and np.array_equal(rng, want["ranges"])
    finally:
        h.set_options(host_sectors=0)
        h.set_view(-180.0, 180.0, lat=LAT, lon=LON)


def test_a_long_series_with_two_in_flight(scene):
    """twelve panoramas, begin k+1 before end k, views changing: whatever was issued from inside another panorama's end (the
    next one's copies) lands in the right buffers"""
    h, od, W, H = scene
    if h.options()["host_dense"]:
        return
    views = [(-180.0, 180.0, 40000.0 + 9000.0 * (k % 3), LAT + 1e-3 * (k % 4), LON - 1e-3 * (k % 3)) for k in range(12)]
    bufs = [(np.empty((H, W, 3), np.uint8), np.empty((H, W), np.float32)) for _ in range(2)]
    got = []
    def begin(k):
        az0, az1, zfar, lat, lon = views[k]
        h.set_view(az0, az1, lat=lat, lon=lon, zfar=zfar)
        h.render_begin(*bufs[k % 2])
    begin(0)
    for k in range(1, len(views) + 1):
        if k < len(views):
            begin(k)
        h.render_end()
        got.append((bufs[(k - 1) % 2][0].copy(), bufs[(k - 1) % 2][1].copy()))
    cache = {}
    for k, v in enumerate(views):
        if v not in cache:
            az0, az1, zfar, lat, lon = v
            cache[v] = _want(od, W, H, az0, az1, zfar, lat, lon)
        assert np.array_equal(got[k][0], cache[v]["bgr"]) and np.array_equal(got[k][1], cache[v]["ranges"]), k
    h.set_view(-180.0, 180.0, lat=LAT, lon=LON)


@pytest.mark.parametrize("sectors", [0, 3])
def test_a_panorama_without_any_terrain(scene, sectors):
    """a viewer 9 km up with a far clip of 3 km: what is that near lies below the frame - no blob is sent, every sector's stream is
    empty, the caller's buffers are sky all over (reference horizonator-lib.c:185, :1016) - in a single call and with two in flight"""
    h, od, W, H = scene
    if h.options()["host_dense"]:
        return
    h.set_options(host_sectors=sectors)
    lib, ctx = h._lib, C.byref(h._ctx)
    try:
        v = od.view(LAT, LON, W, H, -180.0, 180.0, zfar=3000.0, viewer_z=9000.0)
        want = oracle.render(od.mosaic(), v, W, H)
        assert not (want["index"] >= 0).any(), "the scene of this test is meant to be empty"
        h.set_view(-180.0, 180.0, zfar=3000.0)
        z = C.c_float(9000.0)
        assert lib.horizonator_move(ctx, C.byref(z), LAT, LON)
        bufs = [(np.full((H, W, 3), 7, np.uint8), np.full((H, W), 7.0, np.float32)) for _ in range(2)]
        assert lib.horizonator_render_offscreen(ctx, bufs[0][0].ctypes.data, bufs[0][1].ctypes.data)
        assert np.array_equal(bufs[0][0], want["bgr"]) and np.array_equal(bufs[0][1], want["ranges"])
        bufs[0][0][:] = 9; bufs[0][1][:] = 9.0
        for b in bufs:
            assert lib.horizonator_amd_render_begin(ctx, b[0].ctypes.data, b[1].ctypes.data)
        assert lib.horizonator_amd_render_end(ctx) and lib.horizonator_amd_render_end(ctx)
        for b in bufs:
            assert np.array_equal(b[0], want["bgr"]) and np.array_equal(b[1], want["ranges"])
    finally:
        h.set_options(host_sectors=0)
        assert lib.horizonator_move(ctx, None, LAT, LON)
        h.set_view(-180.0, 180.0, zfar=40000.0)
