"""annotator passes over the range image ("next" row N2)"""
import ctypes as C

import numpy as np
import pytest

import hzutil
import oracle
from horizonator_amd import _lib

LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON


def test_host_project_unproject_agree_with_the_oracle_restatement():
    """include/horizonator.h's pure-math functions (product, C) against the
    oracle's separate restatement of reference horizonator-lib.c:1053-1213"""
    lib, orc = _lib.load(), oracle.load()
    rng = np.random.default_rng(5)
    W, H = 4000, 1000
    for _ in range(300):
        lat, lon = LAT + rng.uniform(-0.5, 0.5), LON + rng.uniform(-0.5, 0.5)
        ele = rng.uniform(0, 3000)
        # not exactly 360 degrees: reference horizonator_x_from_az collapses an
        # exact 2*pi span to zero (C round() of 0.5), see DESIGN.md
        az0, az1 = np.deg2rad(-179.9), np.deg2rad(179.95)
        a = [C.c_double() for _ in range(3)]
        b = [C.c_double() for _ in range(3)]
        cosl = np.cos(np.deg2rad(LAT))
        ra = lib.horizonator_project(*[C.byref(v) for v in a], LAT, cosl, LON, 1000.0, lat, lon, ele, az0, az1, W, H)
        rb = orc.orc_project(*[C.byref(v) for v in b], LAT, cosl, LON, 1000.0, lat, lon, ele, az0, az1, W, H)
        assert bool(ra) == bool(rb)
        if ra:
            assert [v.value for v in a] == [v.value for v in b]
        x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
        la, lo, lb, lob = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        rr = rng.uniform(200, 50000)
        assert lib.horizonator_unproject(C.byref(la), C.byref(lo), x, y, rr, -1.0, LAT, cosl, LON, -180.0, 180.0, W, H)
        assert orc.orc_unproject(C.byref(lb), C.byref(lob), x, y, rr, -1.0, LAT, cosl, LON, -180.0, 180.0, W, H)
        assert (la.value, lo.value) == (lb.value, lob.value)
        assert not lib.horizonator_unproject(C.byref(la), C.byref(lo), x, y, rr, rr, LAT, cosl, LON, -180.0, 180.0, W, H)


def test_project_then_unproject_round_trip():
    orc = oracle.load()
    W, H = 8000, 2000
    cosl = np.cos(np.deg2rad(LAT))
    x, y, r = C.c_double(), C.c_double(), C.c_double()
    lat, lon, ele = LAT + 0.11, LON - 0.07, 1800.0
    assert orc.orc_project(C.byref(x), C.byref(y), C.byref(r), LAT, cosl, LON, 900.0, lat, lon, ele,
                           np.deg2rad(-179.9), np.deg2rad(179.95), W, H)
    la, lo = C.c_float(), C.c_float()
    assert orc.orc_unproject(C.byref(la), C.byref(lo), int(round(x.value)), int(round(y.value)), r.value, -1.0,
                             LAT, cosl, LON, -179.9, 179.95, W, H)
    assert abs(la.value - lat) < 2e-4 and abs(lo.value - lon) < 2e-4        # within a pixel's worth


@pytest.mark.gpu
def test_device_passes_match_the_oracle():
    import horizonator_amd
    R, W, H = 400, 2000, 500
    d = hzutil.dem_dir_for(LAT, LON, R)
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    AZ0, AZ1 = -179.9, 179.95          # the CLI never asks for exactly 360 degrees either
    _, ranges = h.render(AZ0, AZ1, zfar=60000.0)
    lat32, lon32 = float(np.float32(LAT)), float(np.float32(LON))

    # link cells (reference annotator.c:228-264): the same bits as the oracle's restatement of that loop - the
    # transcendental functions are the host's (glibc, as in the reference), the device multiplies and divides
    for cut in (0, 37):
        la, lo = h.link_cells(14, 14, cut)
        ola, olo = oracle.link_cells(ranges, 14, 14, cut, lat32, lon32, float(np.float32(AZ0)), float(np.float32(AZ1)))
        assert la.shape == ola.shape and la.size > 1000
        assert np.array_equal(la, ola, equal_nan=True) and np.array_equal(lo, olo, equal_nan=True)
        assert (~np.isnan(ola)).sum() > 300

    # points of interest: every 9th cell's ground point (visible by construction,
    # when inside the distance window), the same points 400 m underground and
    # 2 km up in the air, and random points
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    v = h.view()
    rng = np.random.default_rng(3)
    jj, ii = np.meshgrid(np.arange(5, 2 * R - 5, 9), np.arange(5, 2 * R - 5, 9), indexing="ij")
    glat = LAT + (jj - v["viewer_cell_j"]) / 1200.0
    glon = LON + (ii - v["viewer_cell_i"]) / 1200.0
    gele = m[jj, ii].astype(np.float32)
    pois = np.concatenate([
        np.stack([glat.ravel(), glon.ravel(), gele.ravel()], 1),
        np.stack([glat.ravel(), glon.ravel(), gele.ravel() - 400], 1),
        np.stack([glat.ravel(), glon.ravel(), gele.ravel() + 2000], 1),
        np.stack([LAT + rng.uniform(-1, 1, 500), LON + rng.uniform(-1, 1, 500), rng.uniform(0, 3000, 500)], 1),
    ]).astype(np.float32)
    vis, x, y = h.poi_visibility(pois, cut_off_bottom_px=20)
    ovis, ox, oy = oracle.poi_visibility(ranges, pois, 20, lat32, lon32, float(np.float32(AZ0)), float(np.float32(AZ1)),
                                           float(np.float32(v["viewer_z"])))
    assert 0.02 < ovis.mean() < 0.9
    # exact: the projection is the host's (glibc atan2 / sqrt, as in the reference), the device searches the range image
    assert np.array_equal(vis, ovis) and np.array_equal(x, ox) and np.array_equal(y, oy)
    h.close()
