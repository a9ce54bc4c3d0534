"""R2 (uniform derivation) and R8 (depth -> range) against a third restatement (tests/naive_host_math.py) that shares
no code with the oracle or with the product: reference horizonator-lib.c:765-799, 864-885, 1006-1047 and the window
arithmetic of dem.c:139-152.  The reference's own horizonator-lib.c cannot be compiled here (GL headers), so these
host formulas are otherwise only ever compared between the oracle's C restatement and the product's C."""
import ctypes as C

import numpy as np
import pytest

import hzutil
import naive_host_math as nv
import oracle

# viewpoints on all four sides of the equator and the prime meridian (negative tile numbers, tile names S../W..)
PLACES = ((34.4137, -117.5621), (-33.9321, 18.4317), (46.5593, 8.0414), (-13.1631, -72.5450), (0.3127, -0.4211))


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    for k in range(n):
        lat0, lon0 = PLACES[k % len(PLACES)]
        R = int(rng.choice([24, 60, 130]))
        frac = R / 1200.0 * 0.7
        lat, lon = lat0 + rng.uniform(-frac, frac), lon0 + rng.uniform(-frac, frac)
        W, H = int(rng.integers(16, 3000)), int(rng.integers(5, 1200))
        if k % 3 == 0:
            H |= 1                                                                  # odd heights: the middle row
        span = float(rng.choice([360.0, 400.0, 725.5, rng.uniform(0.5, 20.0), rng.uniform(20.0, 359.0)]))
        az0 = float(rng.uniform(-720.0, 720.0))
        vz = None if k % 4 else float(rng.uniform(0.0, 9000.0))
        znear, zfar = float(rng.choice([1.0, 100.0, 500.0])), float(rng.choice([2000.0, 40000.0, 600000.0]))
        yield dict(lat0=lat0, lon0=lon0, R=R, lat=lat, lon=lon, W=W, H=H, az0=az0, az1=az0 + span, viewer_z=vz, znear=znear, zfar=zfar)


def _naive_view(c, od, mosaic):
    """the naive uniforms for case c; the window is that of a context made at (lat0, lon0)"""
    tile, cell = nv.window(c["lat0"], c["lon0"], c["R"])
    assert tile == list(od.d.origin_tile) and cell == list(od.d.origin_cell), "window arithmetic (dem.c:139-152)"
    return nv.move(c["lat"], c["lon"], tile, cell, lambda i, j: mosaic[j, i], c["viewer_z"])


def test_oracle_uniforms_and_tanel_equal_the_naive_restatement():
    n = 0
    doms = {}
    for c in _cases(1000, 77):
        key = (c["lat0"], c["lon0"], c["R"])
        if key not in doms:
            od = oracle.Dem(c["lat0"], c["lon0"], hzutil.dem_dir_for(c["lat0"], c["lon0"], c["R"]), radius_cells=c["R"])
            doms[key] = (od, od.mosaic())
        od, mosaic = doms[key]
        want = _naive_view(c, od, mosaic)
        v = od.view(c["lat"], c["lon"], c["W"], c["H"], c["az0"], c["az1"], viewer_z=-1.0 if c["viewer_z"] is None else c["viewer_z"],
                    znear=c["znear"], zfar=c["zfar"])
        for k, x in want.items():
            assert np.float32(getattr(v, k)) == x, (k, c)
        assert np.float32(v.aspect) == np.float32(c["W"]) / np.float32(c["H"])
        # tan(elevation) per GL row, including the mirrored upper half (reference :1026-1047)
        t = oracle.tanel(c["W"], c["H"], v.az_deg0, v.az_deg1)
        H = c["H"]
        rows = sorted(set([0, 1, H // 2 - 1, H // 2, (H - 1) // 2, H - H // 2, H - 2, H - 1]) & set(range(H)))
        for row in rows:
            y = row if row < H - H // 2 else H - 1 - row
            assert t[row] == nv.get_tanel(y, c["W"], H, v.az_deg0, v.az_deg1), (row, c)
        n += 1
    assert n == 1000


def test_oracle_ranges_equal_the_naive_loop():
    """the oracle's depth -> range conversion of random depth images, odd and even heights, against the naive loop with
    glibc's hypotf (the reference's own call, :1024)"""
    lib = oracle.load()
    rng = np.random.default_rng(5)
    for k in range(24):
        W, H = int(rng.integers(3, 40)), int(rng.integers(1, 31))
        az0 = float(rng.uniform(-400, 400)); az1 = az0 + float(rng.choice([360.0, 11.5, 170.0, 400.0]))
        znear, zfar = float(rng.choice([1.0, 100.0])), float(rng.choice([2000.0, 40000.0, 600000.0]))
        z24 = rng.integers(0, 1 << 24, size=(H, W), dtype=np.uint32)
        z24[rng.random((H, W)) < 0.3] = 0xFFFFFF
        z24[0, 0] = 0; z24[-1, -1] = 0xFFFFFE
        want = nv.ranges_from_depth(z24, W, H, az0, az1, znear, zfar)
        got = np.empty((H, W), np.float32)
        lib.orc_ranges_from_z24.restype = None
        lib.orc_ranges_from_z24(C.c_void_p(got.ctypes.data), C.c_void_p(z24.ctypes.data), W, H,
                                C.c_float(az0), C.c_float(az1), C.c_float(znear), C.c_float(zfar))
        assert np.array_equal(got, want), (k, W, H)


@pytest.mark.gpu
def test_product_uniforms_and_ranges_equal_the_naive_restatement():
    """the PRODUCT's horizonator_move / pan_zoom / set_zextents (hz_host.c) and its depth -> range conversion (device
    kernels + hz_scatter.c) against the naive restatement: 1000 moves over contexts on both sides of the equator and the
    prime meridian, and whole renders (odd and even heights, spans of 360 degrees and more)"""
    import horizonator_amd
    ctxs = {}
    n = 0
    for c in _cases(1000, 78):
        key = (c["lat0"], c["lon0"], c["R"])
        if key not in ctxs:
            d = hzutil.dem_dir_for(c["lat0"], c["lon0"], c["R"])
            h = horizonator_amd.horizonator(c["lat0"], c["lon0"], 64, 32, dir_dems=d, render_radius_cells=c["R"])
            od = oracle.Dem(c["lat0"], c["lon0"], d, radius_cells=c["R"])
            ctxs[key] = (h, od, od.mosaic())
        h, od, mosaic = ctxs[key]
        want = _naive_view(c, od, mosaic)
        ctx, lib = C.byref(h._ctx), h._lib
        vz = C.c_float(-1.0 if c["viewer_z"] is None else c["viewer_z"])
        assert lib.horizonator_pan_zoom(ctx, c["az0"], c["az1"])
        assert lib.horizonator_move(ctx, C.byref(vz) if n % 2 else (None if c["viewer_z"] is None else C.byref(vz)), c["lat"], c["lon"])
        assert lib.horizonator_set_zextents(ctx, c["znear"], c["zfar"], c["znear"], c["zfar"])
        assert np.float32(vz.value) == want["viewer_z"] or (c["viewer_z"] is None and not n % 2)      # (reported back through the in/out pointer)
        v = h.view()
        for k, x in want.items():
            assert np.float32(v[k]) == x, (k, c)
        assert np.float32(v["az_deg0"]) == np.float32(c["az0"]) and np.float32(v["az_deg1"]) == np.float32(c["az1"])
        assert np.float32(v["znear"]) == np.float32(c["znear"]) and np.float32(v["zfar"]) == np.float32(c["zfar"])
        n += 1
    assert n == 1000
    for h, od, _ in ctxs.values():
        h.close()
    # whole renders: ranges == naive(z24)
    rng = np.random.default_rng(9)
    for k in range(10):
        lat0, lon0 = PLACES[k % len(PLACES)]
        R = 40
        W, H = int(rng.integers(20, 120)), int(rng.integers(9, 60)) | (k & 1)
        d = hzutil.dem_dir_for(lat0, lon0, R)
        h = horizonator_amd.horizonator(lat0, lon0, W, H, dir_dems=d, render_radius_cells=R)
        az0 = float(rng.uniform(-400, 400)); az1 = az0 + float([360.0, 33.0, 400.0, 170.0][k % 4])
        zfar = float([4000.0, 40000.0][k % 2])
        _, ranges, _, z24 = h.render_full(az0, az1, zfar=zfar)
        v = h.view()
        want = nv.ranges_from_depth(z24[::-1], W, H, v["az_deg0"], v["az_deg1"], v["znear"], v["zfar"])
        assert np.array_equal(ranges, want), (k, W, H, az0, az1)
        assert (z24 != 0xFFFFFF).any()
        h.close()
