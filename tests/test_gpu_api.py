"""The reference's API surface (include/horizonator.h and the Python mirror of
horizonator-pywrap.c) on the GPU: same behaviour, results equal to the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import hzutil
import oracle

pytestmark = pytest.mark.gpu

LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON


@pytest.fixture(scope="module")
def scene():
    import horizonator_amd
    R, W, H = 300, 1200, 300
    d = hzutil.dem_dir_for(LAT, LON, R)
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    yield h, od, W, H
    h.close()


def test_device_mosaic_is_the_dem_window(scene):
    h, od, W, H = scene
    assert np.array_equal(h.mosaic(), od.mosaic())
    assert h.Ntriangles == 2 * (2 * 300 - 1) ** 2


def test_render_returns_reference_shapes_and_matches_oracle(scene):
    h, od, W, H = scene
    image, ranges = h.render(-180, 180)
    assert image.shape == (H, W, 3) and image.dtype == np.uint8
    assert ranges.shape == (H, W) and ranges.dtype == np.float32
    ref = oracle.render(od.mosaic(), od.view(LAT, LON, W, H, -180, 180), W, H)
    assert np.array_equal(image, ref["bgr"]) and np.array_equal(ranges, ref["ranges"])
    # the four return combinations of reference horizonator-pywrap.c:198-202,262-270
    assert h.render(-180, 180, return_image=False, return_range=False) == ()
    only_img = h.render(-180, 180, return_range=False)
    only_rng = h.render(-180, 180, return_image=False)
    assert np.array_equal(only_img, image) and np.array_equal(only_rng, ranges)
    # uniforms the host code derived == the oracle's restatement of the reference's host code
    v, ov = h.view(), od.view(LAT, LON, W, H, -180, 180).as_dict()
    assert {k: np.float32(x) for k, x in v.items()} == {k: np.float32(x) for k, x in ov.items()}


def test_render_recycles_only_the_results_the_caller_dropped(scene):
    """render() makes its arrays of memory the caller has let go of (horizonator_amd._ResultMemory; the reference's wrapper
    allocates per call, horizonator-pywrap.c:234-250): results still held stay what they were, whatever is rendered after"""
    h, od, W, H = scene
    first = h.render(-180, 180)
    keep = (first[0].copy(), first[1].copy())
    other = h.render(-40, 100)                          # `first` is held: not its memory
    assert not np.shares_memory(other[0], first[0]) and not np.shares_memory(other[1], first[1])
    assert np.array_equal(first[0], keep[0]) and np.array_equal(first[1], keep[1])
    where = (first[0].ctypes.data, first[1].ctypes.data)
    del first
    again = h.render(-180, 180)                         # ... dropped: the same memory, the same picture
    assert (again[0].ctypes.data, again[1].ctypes.data) == where
    assert np.array_equal(again[0], keep[0]) and np.array_equal(again[1], keep[1])
    ref = oracle.render(od.mosaic(), od.view(LAT, LON, W, H, -40, 100), W, H)
    assert np.array_equal(other[0], ref["bgr"]) and np.array_equal(other[1], ref["ranges"])
    full = h.render_full(-180, 180)
    assert np.array_equal(full[0], keep[0]) and np.array_equal(full[1], keep[1]) and np.array_equal(again[0], keep[0])


def test_move_zextents_and_pixel_centre_azimuths(scene):
    h, od, W, H = scene
    lat, lon = LAT + 0.02, LON - 0.015
    image, ranges, index, z24 = h.render_full(-30, 80, lat=lat, lon=lon, znear=200.0, zfar=15000.0,
                                              znear_color=500.0, zfar_color=9000.0)
    v = od.view(lat, lon, W, H, -30, 80, znear=200.0, zfar=15000.0, znear_color=500.0, zfar_color=9000.0)
    ref = oracle.render(od.mosaic(), v, W, H)
    hzutil.assert_same_render(dict(bgr=image, ranges=ranges, index=index, z24=z24), ref, "moved")
    # az_extents_use_pixel_centers (reference horizonator-pywrap.c:204-212)
    image2, _ = h.render(-30, 80, lat=lat, lon=lon, az_extents_use_pixel_centers=True)
    step = (80 - -30) / (W - 1)
    v2 = od.view(lat, lon, W, H, -30 - step / 2, 80 + step / 2)
    assert np.array_equal(image2, oracle.render(od.mosaic(), v2, W, H)["bgr"])
    # set_zextents refuses non-positive values and leaves the view alone
    lib = h._lib
    before = h.view()
    assert not lib.horizonator_set_zextents(C.byref(h._ctx), -1.0, 100.0, 1.0, 1.0)
    assert h.view() == before


def test_pick_inverts_the_projection(scene):
    h, od, W, H = scene
    _, ranges, index, z24 = h.render_full(-180, 180, lat=LAT, lon=LON)
    ys, xs = np.nonzero(index >= 0)
    k = len(ys) // 2
    x, y = int(xs[k]), int(ys[k])
    got = h.pick(x, y)
    assert got is not None
    # reference horizonator-lib.c:1285-1295: unproject the depth as a horizontal distance
    # ... the expectation from the ORACLE's restatement of it (oracle_annot.c: orc_unproject), not from the library under test
    lat, lon = C.c_float(), C.c_float()
    v = h.view()
    depth = np.float32(np.float64(z24[y, x]) * (1.0 / 16777215.0))
    range_en = float(depth * np.float32(v["zfar"] - v["znear"]) + np.float32(v["znear"]))
    assert oracle.load().orc_unproject(C.byref(lat), C.byref(lon), x, y, -1.0, range_en, np.float32(LAT),
                                       v["cos_viewer_lat"], np.float32(LON), v["az_deg0"], v["az_deg1"], W, H)
    assert got == (lat.value, lon.value)
    # (and the library's own horizonator_unproject says the same)
    lat2, lon2 = C.c_float(), C.c_float()
    assert h._lib.horizonator_unproject(C.byref(lat2), C.byref(lon2), x, y, -1.0, range_en, np.float32(LAT),
                                        v["cos_viewer_lat"], np.float32(LON), v["az_deg0"], v["az_deg1"], W, H)
    assert (lat2.value, lon2.value) == (lat.value, lon.value)
    # and the picked point is the visible cell, to within a cell or two
    cell = index[y, x] >> 1
    N = 2 * h.radius_cells
    j, i = divmod(int(cell), N - 1)
    vi = h.view()
    cell_lat = LAT + (j + 0.5 - vi["viewer_cell_j"]) / 1200.0
    cell_lon = LON + (i + 0.5 - vi["viewer_cell_i"]) / 1200.0
    assert abs(got[0] - cell_lat) < 3 / 1200.0 and abs(got[1] - cell_lon) < 3 / 1200.0
    sky = np.argwhere(index < 0)[0]
    assert h.pick(int(sky[1]), int(sky[0])) is None


def test_sector_and_device_buffers(scene):
    import torch
    h, od, W, H = scene
    h.set_view(-180, 180, lat=LAT, lon=LON)
    image, ranges, index, _ = h.render_full(-180, 180)
    h.set_sector(400, 700)
    try:
        dev = torch.device("cuda:0")
        d_img = torch.empty((H, 300, 3), dtype=torch.uint8, device=dev)
        d_rng = torch.empty((H, 300), dtype=torch.float32, device=dev)
        d_idx = torch.empty((H, 300), dtype=torch.int32, device=dev)
        h.render_device(d_img.data_ptr(), d_rng.data_ptr(), d_idx.data_ptr())
        h.sync()
        assert np.array_equal(d_img.cpu().numpy(), image[:, 400:700])
        assert np.array_equal(d_rng.cpu().numpy(), ranges[:, 400:700])
        assert np.array_equal(d_idx.cpu().numpy(), index[:, 400:700])
    finally:
        h.set_sector(0, W)


def test_errors_behave_like_the_reference(tmp_path):
    import horizonator_amd
    d = hzutil.dem_dir_for(LAT, LON, 32)
    # both radii (reference horizonator-pywrap.c:100-104)
    with pytest.raises(RuntimeError):
        horizonator_amd.horizonator(LAT, LON, 64, 16, dir_dems=d, render_radius_cells=32, render_radius_m=1000.0)
    # texture path: map tiles missing on disk and no downloads -> init fails (reference
    # horizonator-lib.c:284-289); window modes are not part of this build
    with pytest.raises(RuntimeError):
        horizonator_amd.horizonator(LAT, LON, 64, 16, dir_dems=d, render_radius_cells=32, render_texture=True,
                                    dir_tiles=str(tmp_path / "no_tiles_here"), allow_downloads=False)
    from horizonator_amd import _lib
    lib = _lib.load()
    ctx = _lib.Context()
    assert not lib.horizonator_init(C.byref(ctx), LAT, LON, None, -1, -1, 32, -1.0, True, False, False,
                                    d.encode(), None, None, None, True)
    assert not lib.horizonator_init(C.byref(ctx), LAT, LON, None, 64, 16, 32, -1.0, False, False, False,
                                    d.encode(), None, None, None, True)
    # wrong-size tile -> init fails (reference dem.c:234-239)
    bad = tmp_path / "bad"
    bad.mkdir()
    (bad / "N34W118.hgt").write_bytes(b"\0" * 10)
    with pytest.raises(RuntimeError):
        horizonator_amd.horizonator(LAT, LON, 64, 16, dir_dems=str(bad), render_radius_cells=32)
    # a context that was deinit'ed refuses everything
    h = horizonator_amd.horizonator(LAT, LON, 64, 16, dir_dems=d, render_radius_cells=32)
    ctxp = C.byref(h._ctx)
    lib.horizonator_deinit(ctxp)
    assert not lib.horizonator_pan_zoom(ctxp, 0.0, 10.0)
    assert not lib.horizonator_render_offscreen(ctxp, None, None)
    h._ctx = None
    # viewer_z in/out (reference horizonator.h:90-93)
    ctx = _lib.Context()
    z = C.c_float(-1.0)
    assert lib.horizonator_init(C.byref(ctx), LAT, LON, C.byref(z), 64, 16, 32, -1.0, True, False, False,
                                d.encode(), None, None, None, True)
    od = oracle.Dem(LAT, LON, d, radius_cells=32)
    assert z.value == od.view(LAT, LON, 64, 16, 0, 1).viewer_z
    assert not lib.horizonator_resized(C.byref(ctx), 10, 10)
    lib.horizonator_deinit(C.byref(ctx))


@pytest.mark.parametrize("how", ["device", "host", None])
def test_either_ingest_builds_the_same_mosaic(monkeypatch, how):
    """the DEM's tiles decoded by k_ingest (hz_ingest.cpp: the default since round 6) or on the host (HORIZONATOR_INGEST=host,
    round 1's way): the window of reference dem.c:264-309 either way - a window of 2x2 tiles (through the public dem context)
    and one of 5x5 (the library's own tile table, staged through pinned memory in several pieces)"""
    import horizonator_amd
    for R, W, H in ((700, 64, 16), (2100, 4000, 1000)):
        d = hzutil.dem_dir_for(LAT, LON, R)
        if how is None:
            monkeypatch.delenv("HORIZONATOR_INGEST", raising=False)
        else:
            monkeypatch.setenv("HORIZONATOR_INGEST", how)
        h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
        assert np.array_equal(h.mosaic(), oracle.Dem(LAT, LON, d, radius_cells=R).mosaic()), (how, R)
        h.close()


def test_more_tiles_than_the_reference_can_load():
    """5x5 tiles: the reference refuses (reference dem.h:8); same semantics here"""
    import horizonator_amd
    R = 2100
    d = hzutil.dem_dir_for(LAT, LON, R)
    h = horizonator_amd.horizonator(LAT, LON, 512, 128, dir_dems=d, render_radius_cells=R)
    assert list(h._ctx.dems.Ndems_ij) == [5, 5]
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    assert np.array_equal(h.mosaic(), od.mosaic())
    image, ranges, index, z24 = h.render_full(-180, 180, zfar=300000.0)
    ref = oracle.render(od.mosaic(), od.view(LAT, LON, 512, 128, -180, 180, zfar=300000.0), 512, 128)
    hzutil.assert_same_render(dict(bgr=image, ranges=ranges, index=index, z24=z24), ref, "5x5")
    h.close()


def test_viewpoint_batch_equals_one_render_per_viewpoint(scene):
    """BASELINE.json configs[3] in small: horizonator_amd_render_batch() == move + render
    per viewpoint == the oracle, and reports the viewer heights it stood at"""
    import torch
    h, od, W, H = scene
    n = 9
    lats = np.array([LAT + 0.03 * (v // 3 - 1) for v in range(n)], np.float32)
    lons = np.array([LON + 0.04 * (v % 3 - 1) for v in range(n)], np.float32)
    h.set_view(-180, 180, zfar=30000.0)
    d_img = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda:0")
    d_rng = torch.empty((n, H, W), dtype=torch.float32, device="cuda:0")
    z = h.render_batch(lats, lons, d_img.data_ptr(), d_rng.data_ptr())
    h.sync()
    img, rng = d_img.cpu().numpy(), d_rng.cpu().numpy()
    mosaic = od.mosaic()
    for v in range(n):
        ov = od.view(float(lats[v]), float(lons[v]), W, H, -180, 180, zfar=30000.0)
        assert np.float32(ov.viewer_z) == z[v]
        ref = oracle.render(mosaic, ov, W, H, want=("bgr", "ranges"))
        assert np.array_equal(img[v], ref["bgr"]) and np.array_equal(rng[v], ref["ranges"]), v
    # ... and equals the one-at-a-time API of the reference
    one_img, one_rng = h.render(-180, 180, lat=float(lats[4]), lon=float(lons[4]), zfar=30000.0)
    assert np.array_equal(one_img, img[4]) and np.array_equal(one_rng, rng[4])
    # explicit viewer heights are taken as given
    z2 = h.render_batch(lats[:2], lons[:2], d_img.data_ptr(), 0, viewer_z=[2500.0, 2600.0])
    h.sync()
    assert list(z2) == [2500.0, 2600.0]
    ov = od.view(float(lats[1]), float(lons[1]), W, H, -180, 180, viewer_z=2600.0, zfar=30000.0)
    assert np.array_equal(d_img[1].cpu().numpy(), oracle.render(mosaic, ov, W, H, want=("bgr",))["bgr"])


def test_context_from_a_broadcast_mosaic(scene):
    """SURVEY.md 8e DEM distribution: a context made from another context's window + mosaic
    (no tile read) renders the same bytes, also after moves with automatic viewer height"""
    import horizonator_amd
    h, od, W, H = scene
    h2 = horizonator_amd.horizonator.from_mosaic(LAT, LON, W, H, h.window(), h.mosaic())
    try:
        assert h2.window() == h.window() and np.array_equal(h2.mosaic(), h.mosaic())
        assert h2.Ntriangles == h.Ntriangles and h2.radius_cells == h.radius_cells
        for lat, lon in [(LAT, LON), (LAT + 0.031, LON - 0.027)]:
            a = h.render_full(-180, 180, lat=lat, lon=lon, zfar=20000.0)
            b = h2.render_full(-180, 180, lat=lat, lon=lon, zfar=20000.0)
            assert h2.view() == h.view()
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
        assert h2.texture_layout() == h.texture_layout()
        with pytest.raises(ValueError):
            horizonator_amd.horizonator.from_mosaic(LAT, LON, W, H, h.window(), h.mosaic()[:-1])
    finally:
        h2.close()


def test_pipelined_renders_equal_waited_for_renders():
    """The library overlaps consecutive renders (three framebuffers and queue sets in turn, four streams).
    A seeded random sequence of moves, azimuth / depth-extent / sector changes, texture switches,
    picks and renders into separate device buffers, queued WITHOUT waiting in between, must leave
    exactly what the same sequence leaves when every render is waited for."""
    import torch
    import horizonator_amd
    R, W, H = 150, 700, 180
    d = hzutil.dem_dir_for(LAT, LON, R)
    texels = None

    def run(wait_every_time):
        nonlocal texels
        rng = np.random.default_rng(77)
        h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
        outs, picks = [], []
        try:
            if texels is None:
                _, _, nx, ny = h.texture_layout()
                texels = hzutil.hash_texture(ny * 256, nx * 256, seed=3, blocky=4)
            c0, c1 = 0, W
            for step in range(40):
                op = int(rng.integers(0, 7))
                if op == 0:
                    az0 = float(rng.uniform(-400, 400))
                    h.set_view(az0, az0 + float(rng.choice([360.0, 90.0, 17.5])), zfar=float(rng.choice([8000.0, 30000.0])))
                elif op == 1:
                    h.set_view(-180, 180, lat=LAT + float(rng.uniform(-0.05, 0.05)), lon=LON + float(rng.uniform(-0.05, 0.05)),
                               zfar=20000.0)
                elif op == 2:
                    c0 = int(rng.integers(0, W - 50)); c1 = int(rng.integers(c0 + 20, W + 1))
                    h.set_sector(c0, c1)
                elif op == 3:
                    h.set_texture(texels if rng.integers(0, 2) else None)
                elif op == 4 and outs:
                    picks.append(h.pick(int(rng.integers(c0, c1)), int(rng.integers(0, H))))
                # every step renders
                img = torch.zeros((H, c1 - c0, 3), dtype=torch.uint8, device="cuda:0")
                rng_ = torch.zeros((H, c1 - c0), dtype=torch.float32, device="cuda:0")
                h.render_device(img.data_ptr(), rng_.data_ptr())
                if wait_every_time:
                    h.sync()
                outs.append((img, rng_))
            h.sync()
            return [(a.cpu().numpy(), b.cpu().numpy()) for a, b in outs], picks
        finally:
            h.close()

    waited, picks_w = run(True)
    queued, picks_q = run(False)
    assert len(waited) == len(queued) == 40
    for k, ((ia, ra), (ib, rb)) in enumerate(zip(waited, queued)):
        assert np.array_equal(ia, ib) and np.array_equal(ra, rb), k
    assert picks_w == picks_q


def test_contexts_from_several_threads():
    """contexts are created, used and destroyed concurrently from different threads (one thread
    per context): same bytes as the serial render"""
    import threading
    import horizonator_amd
    R, W, H = 64, 320, 80
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    ref = oracle.render(od.mosaic(), od.view(LAT, LON, W, H, -180, 180, zfar=9000.0), W, H, want=("bgr", "ranges"))
    results, errors = [None] * 6, []

    def work(k):
        try:
            for _ in range(3):
                h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
                try:
                    results[k] = h.render(-180, 180, zfar=9000.0)
                finally:
                    h.close()
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for image, ranges in results:
        assert np.array_equal(image, ref["bgr"]) and np.array_equal(ranges, ref["ranges"])


@pytest.mark.parametrize("two_pass", ["1", "0"])
def test_single_renders_through_the_api_with_the_round_count_forced(two_pass, monkeypatch):
    """One render at a time, each converted at once, through horizonator.h's calls - with two
    rounds forced on a scene far below the size where they are chosen, so that the first round is
    the longer one: a draw that finds the chip idle runs its second round beside its first, and
    the conversion has to wait for both (it once waited for the second only).  Repeated, on a
    fresh and on a used context, with moves in between: every output against the oracle."""
    import horizonator_amd
    monkeypatch.setenv("HZ_TWO_PASS", two_pass)
    R, W, H = 300, 1200, 300
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    used = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    try:
        views = [(LAT, LON, -180, 180, 20000.0), (LAT + 0.031, LON - 0.027, -180, 180, 20000.0), (LAT - 0.02, LON + 0.015, -60, 95, 60000.0)]
        want = []
        for lat, lon, a0, a1, zfar in views:
            v = od.view(lat, lon, W, H, a0, a1, zfar=zfar)
            want.append(oracle.render(m, v, W, H))
        for rep in range(6):
            fresh = horizonator_amd.horizonator.from_mosaic(LAT, LON, W, H, used.window(), m)
            try:
                for h in (fresh, used):
                    for (lat, lon, a0, a1, zfar), o in zip(views, want):
                        image, ranges, index, z24 = h.render_full(a0, a1, lat=lat, lon=lon, zfar=zfar)
                        assert np.array_equal(index, o["index"]) and np.array_equal(z24, o["z24"]), (rep, lat, lon)
                        assert np.array_equal(image, o["bgr"]) and np.array_equal(ranges, o["ranges"]), (rep, lat, lon)
            finally:
                fresh.close()
    finally:
        used.close()


def test_which_draws_keep_coarse_depth(monkeypatch):
    """hz_hip_last_plan: a zoomed view keeps coarse depth (hz_k_hiz.h) and its first round reaches as far as the
    cap allows; a whole panorama that is waited for does not (its second round runs beside its first); in a series of
    renders the later ones do (they find the marching kernel of the render before them still running) - and the
    pictures of the series are those of the renders that were waited for."""
    import torch
    import horizonator_amd
    for k in [k for k in os.environ if k.startswith("HZ_") and k != "HZ_TEST_DEM_DIR"]:
        monkeypatch.delenv(k)                       # (tools/gpu_modes.sh runs the suite under switches that change the plan: this test is about the defaults)
    monkeypatch.setenv("HZ_TWO_PASS", "1")          # (two rounds at a size the test can afford)
    R, W, H = 1000, 8000, 2000
    d = hzutil.dem_dir_for(LAT, LON, R)
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    lib = h._lib
    dev = lib.horizonator_amd_device(C.byref(h._ctx))

    def plan():
        out = (C.c_int * 5)()
        assert lib.hz_hip_last_plan(dev, out) == 0
        return [int(x) for x in out][:4]

    try:
        img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
        rng = torch.empty((H, W), dtype=torch.float32, device="cuda:0")
        h.set_view(-180, 180, zfar=200000.0)
        h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
        h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
        rounds, coarse, reach, listed = plan()
        assert (rounds, coarse, listed) == (2, 0, 0) and reach == 64, plan()      # 8000 columns: cells wider than 20 px = 64 cells
        waited = (img.cpu().numpy().copy(), rng.cpu().numpy().copy())
        seen = 0
        for _ in range(12):                         # a series: nobody waits in between
            h.render_device(img.data_ptr(), rng.data_ptr())
            seen += plan()[1]
        h.sync()
        # (whether a draw finds its predecessor still marching is a race between this thread and the device: a slow host
        # - a profiler attached, a loaded box - sees fewer; the bytes below are what must not depend on it)
        assert seen >= 1, f"none of 12 renders queued back to back kept coarse depth"
        assert np.array_equal(img.cpu().numpy(), waited[0]) and np.array_equal(rng.cpu().numpy(), waited[1])
        h.set_view(-10, 10, zfar=200000.0)          # a 20 degree view: ppr = 22900, the reach hits its cap
        h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
        rounds, coarse, reach, listed = plan()
        assert (rounds, coarse, listed) == (2, 1, 1) and reach == 384, plan()     # (the first draw of a view: the short reach; hz_kernels.hip, adapt)
    finally:
        h.close()


def test_the_reach_of_a_zoomed_view_follows_the_draws_before(monkeypatch):
    """hz_kernels.hip, adapt: the first draw of a zoomed view reaches HZ_NEAR_CELLS_WIDE cells; a later draw of the same view
    that finds the second round's queue counters of a draw before it on the host tries the long reach if they were large
    (HZ_ADAPT_HI, here 0: always), keeps it if it paid and goes back for good if not; another view starts short again.
    The bytes never depend on it."""
    import torch
    import horizonator_amd
    for k in [k for k in os.environ if k.startswith("HZ_") and k != "HZ_TEST_DEM_DIR"]:
        monkeypatch.delenv(k)
    monkeypatch.setenv("HZ_TWO_PASS", "1")
    monkeypatch.setenv("HZ_ADAPT_HI", "0")
    R, W, H = 1000, 8000, 2000
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=hzutil.dem_dir_for(LAT, LON, R), render_radius_cells=R)
    try:
        img = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
        rng = torch.empty((H, W), dtype=torch.float32, device="cuda:0")

        def draw():
            h.render_device(img.data_ptr(), rng.data_ptr()); h.sync()
            return h.last_plan()["reach_cells"], img.cpu().numpy().copy(), rng.cpu().numpy().copy()

        h.set_view(-10, 10, zfar=200000.0)
        reach, img0, rng0 = draw()
        assert reach == 384
        reaches = []
        for _ in range(5):
            reach, i, r = draw()
            reaches.append(reach)
            assert np.array_equal(i, img0) and np.array_equal(r, rng0)
        assert reaches[0] == 512                        # the first draw's counters were there: the long reach is tried
        assert len(set(reaches[2:])) == 1, reaches      # ... and kept, or given up for good
        h.set_view(-12, 8, zfar=200000.0)               # another view: short again
        assert draw()[0] == 384
    finally:
        h.close()


@pytest.mark.parametrize("fast_math", [1, 0])
def test_vertex_cache_draws_the_same_bytes(fast_math):
    """hz_options_t::vertex_cache: the first draw from a viewpoint computes every vertex, the second fills the cache (16 bytes
    per vertex: the view-independent half of the transform, reference vertex.glsl:133-134, 154, 156), the draws after it read
    it - whatever the azimuths, the aspect, the depth and colour extents; a move starts over.  Every one of them against the oracle."""
    import horizonator_amd
    R, W, H = 300, 1600, 400
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    try:
        h.set_options(fast_math=fast_math, vertex_cache=1)
        used = []
        for lat, lon in ((LAT, LON), (LAT + 0.031, LON - 0.017), (LAT, LON)):
            for az0, az1, kw in ((-180.0, 180.0, dict(zfar=90000.0)), (-180.0, 180.0, dict(zfar=90000.0)), (10.0, 75.0, dict(zfar=30000.0)),
                                 (-180.0, 180.0, dict(znear=50.0, zfar=20000.0, znear_color=2000.0, zfar_color=9000.0)), (100.0, 330.0, dict(zfar=90000.0))):
                want = oracle.render(m, od.view(lat, lon, W, H, az0, az1, **kw), W, H)
                got = h.render_full(az0, az1, lat=lat, lon=lon, **kw)
                hzutil.assert_same_render(dict(bgr=got[0], ranges=got[1], index=got[2], z24=got[3]), want, f"{lat} {lon} {az0} {az1} {kw}")
                used.append(h.last_plan()["vertex_cache"])
        # per viewpoint: cold, fill + cached, cached, cached, cached (a call that is drawn in several sectors - HZ_HOST_SECTORS of
        # tools/gpu_modes.sh; this image is one by default - draws from the viewpoint several times itself)
        if h.options()["host_sectors"] in (0, 1):
            assert used == [False, True, True, True, True] * 3, used
        else:
            assert all(used[1:5]) and all(used[6:10]) and all(used[11:15]), used
        h.set_options(vertex_cache=0)
        got = h.render_full(-180.0, 180.0, lat=LAT, lon=LON, zfar=90000.0)
        assert not h.last_plan()["vertex_cache"]
        assert np.array_equal(got[3], oracle.render(m, od.view(LAT, LON, W, H, -180.0, 180.0, zfar=90000.0), W, H)["z24"])
    finally:
        h.close()
