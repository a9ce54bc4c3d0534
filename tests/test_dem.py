"""DEM window arithmetic and sampling: oracle and product against fixtures made
by the reference's own dem.c (tests/golden/dem_samples.npz, oracle/make_golden.py)"""
import ctypes as C
import os

import numpy as np
import pytest

import hzutil
import oracle
from horizonator_amd import _lib

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dem_samples.npz"))


def _cases():
    for k in range(int(GOLD["ncases"])):
        lat, lon, R, Rm, srtm1 = GOLD[f"c{k}_args"]
        yield k, float(lat), float(lon), int(R), float(Rm), bool(srtm1)


def _dir(lat, lon, R, srtm1):
    return hzutil.dem_dir_for(lat, lon, 3000 if R < 0 else R, srtm1=srtm1)


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: f"c{c[0]}")
def test_oracle_dem_matches_reference_dem_c(case):
    k, lat, lon, R, Rm, srtm1 = case
    od = oracle.Dem(lat, lon, _dir(lat, lon, R, srtm1), radius_cells=R, radius_m=Rm, srtm1=srtm1)
    w = GOLD[f"c{k}_window"]
    got = list(od.d.origin_tile) + list(od.d.origin_cell) + list(od.d.ntiles) + [od.d.radius_cells, od.d.cells_per_deg]
    assert got == list(w)
    ii, jj = GOLD[f"c{k}_ij"]
    z = np.array([od.sample(i, j) for i, j in zip(ii, jj)], np.int16)
    assert np.array_equal(z, GOLD[f"c{k}_z"])


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: f"c{c[0]}")
def test_product_dem_api_matches_reference_dem_c(case):
    k, lat, lon, R, Rm, srtm1 = case
    lib = _lib.load()
    ctx = _lib.DemContext()
    assert lib.horizonator_dem_init(C.byref(ctx), lat, lon, R, Rm, _dir(lat, lon, R, srtm1).encode(), srtm1)
    w = GOLD[f"c{k}_window"]
    got = list(ctx.origin_dem_lon_lat) + list(ctx.origin_dem_cellij) + list(ctx.Ndems_ij) + [ctx.radius_cells, ctx.cells_per_deg]
    assert got == list(w)
    ii, jj = GOLD[f"c{k}_ij"]
    z = np.array([lib.horizonator_dem_sample(C.byref(ctx), int(i), int(j)) for i, j in zip(ii, jj)], np.int16)
    assert np.array_equal(z, GOLD[f"c{k}_z"])
    b = [C.c_float() for _ in range(4)]
    lib.horizonator_dem_bounds_latlon_deg(C.byref(ctx), *[C.byref(x) for x in b])
    assert np.array_equal(np.array([x.value for x in b], np.float32), GOLD[f"c{k}_bounds"])
    lib.horizonator_dem_deinit(C.byref(ctx))


class _Window(C.Structure):
    _fields_ = [("cells_per_deg", C.c_int), ("radius_cells", C.c_int), ("origin_tile", C.c_int * 2),
                ("origin_cell", C.c_int * 2), ("ntiles", C.c_int * 2)]


class _Tileset(C.Structure):
    _fields_ = [("win", _Window), ("tile", C.c_void_p), ("tile_bytes", C.c_void_p), ("tile_fd", C.c_void_p)]


def _product_mosaic(lat, lon, R, d, srtm1=False):
    """the mosaic the product uploads to HBM, built by its host code (no GPU involved)"""
    lib = _lib.load()
    lib.hz_window_compute.restype = C.c_bool
    lib.hz_window_compute.argtypes = [C.POINTER(_Window), C.c_float, C.c_float, C.c_int, C.c_float, C.c_bool]
    lib.hz_tileset_open.restype = C.c_bool
    lib.hz_tileset_open.argtypes = [C.POINTER(_Tileset), C.POINTER(_Window), C.c_char_p]
    lib.hz_tileset_build_mosaic.argtypes = [C.POINTER(_Tileset), C.c_void_p]
    lib.hz_tileset_close.argtypes = [C.POINTER(_Tileset)]
    w, ts = _Window(), _Tileset()
    assert lib.hz_window_compute(C.byref(w), lat, lon, R, -1.0, srtm1)
    assert lib.hz_tileset_open(C.byref(ts), C.byref(w), d.encode())
    N = 2 * w.radius_cells
    m = np.empty((N, N), np.int16)
    lib.hz_tileset_build_mosaic(C.byref(ts), m.ctypes.data)
    lib.hz_tileset_close(C.byref(ts))
    return m


@pytest.mark.parametrize("R", [32, 600, 1801, 2100])
def test_product_mosaic_equals_oracle_sampling(R):
    """2100 cells need 5x5 tiles: beyond the reference's 4x4 limit, same semantics"""
    lat, lon = hzutil.VIEW_LAT, hzutil.VIEW_LON
    d = hzutil.dem_dir_for(lat, lon, R)
    od = oracle.Dem(lat, lon, d, radius_cells=R)
    assert np.array_equal(_product_mosaic(lat, lon, R, d), od.mosaic())


def test_window_on_a_tile_corner_reads_in_bounds():
    """origin cell 0 on both axes: the reference indexes tile -1 here (reference
    dem.c:287-291); both restatements read the first tile's own edge instead"""
    lat, lon = 34.0 + 0.5 / 1200, -118.0 + 0.5 / 1200
    R = 1
    d = hzutil.dem_dir_for(lat, lon, 8)
    od = oracle.Dem(lat, lon, d, radius_cells=R)
    assert list(od.d.origin_cell) == [0, 0]
    m = _product_mosaic(lat, lon, R, d)
    assert np.array_equal(m, od.mosaic())
    gen = _lib.load_demgen()
    tile = np.empty((1201, 1201), np.int16)
    gen.hz_demgen_tile_values(tile.ctypes.data, 34, -118, 1200, 0)
    assert m[0, 0] == tile[1200, 0] and m[1, 1] == tile[1199, 1]


def test_missing_and_empty_tiles_read_as_sea_level(tmp_path):
    lat, lon = hzutil.VIEW_LAT, hzutil.VIEW_LON
    src = hzutil.dem_dir_for(lat, lon, 600)
    d = tmp_path / "dems"
    d.mkdir()
    # window of radius 600 touches N33/N34 x W118/W119: keep one, empty one, drop two
    os.symlink(os.path.join(src, "N34W118.hgt"), d / "N34W118.hgt")
    (d / "N33W118.hgt").write_bytes(b"")
    od = oracle.Dem(lat, lon, str(d), radius_cells=600)
    m = od.mosaic()
    assert (m == 0).any() and (m > 0).any()
    assert np.array_equal(_product_mosaic(lat, lon, 600, str(d)), m)


def test_wrong_size_tile_is_an_error(tmp_path):
    lib = _lib.load()
    d = tmp_path / "dems"
    d.mkdir()
    (d / "N34W118.hgt").write_bytes(b"\0" * 1000)
    ctx = _lib.DemContext()
    assert not lib.horizonator_dem_init(C.byref(ctx), hzutil.VIEW_LAT, hzutil.VIEW_LON, 32, -1.0, str(d).encode(), False)
    with pytest.raises(RuntimeError):
        oracle.Dem(hzutil.VIEW_LAT, hzutil.VIEW_LON, str(d), radius_cells=32)


def test_radius_arguments_are_exclusive():
    lib = _lib.load()
    ctx = _lib.DemContext()
    d = hzutil.dem_dir_for(hzutil.VIEW_LAT, hzutil.VIEW_LON, 32).encode()
    assert not lib.horizonator_dem_init(C.byref(ctx), 34.4, -117.5, -1, -1.0, d, False)
    assert not lib.horizonator_dem_init(C.byref(ctx), 34.4, -117.5, 32, 100.0, d, False)


def test_public_api_refuses_more_than_4x4_tiles():
    """reference dem.c:173-178"""
    lib = _lib.load()
    ctx = _lib.DemContext()
    d = hzutil.dem_dir_for(hzutil.VIEW_LAT, hzutil.VIEW_LON, 2100).encode()
    assert not lib.horizonator_dem_init(C.byref(ctx), hzutil.VIEW_LAT, hzutil.VIEW_LON, 2100, -1.0, d, False)
