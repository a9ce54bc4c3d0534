"""BASELINE.json configs[4]: 11x11 SRTM1 tiles (39600^2 samples, 3.1 G triangles), 32768x8192
panorama - the configuration that exists for the azimuth split over 8 GPUs.  It is beyond what
the reference can load (4x4 tiles: reference dem.h:8; 16-bit vertex coordinates and 32-bit
index counts: reference horizonator-lib.c:423-425,475-476,487-512), so the checker is the
oracle alone - on one 1/16 azimuth sector, every output - plus what does not depend on size:
the 8 sectors tile the single-GPU render byte for byte, both rasterisers agree, two runs agree."""
import numpy as np
import pytest

import hzutil
import oracle

pytestmark = pytest.mark.gpu

LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
R, W, H = 19800, 32768, 8192
ZFAR = 600000.0


@pytest.fixture(scope="module")
def cfg5():
    import torch
    import horizonator_amd
    dems = hzutil.dem_dir_for(LAT, LON, R, srtm1=True)
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R, SRTM1=True)
    h.set_view(-180.0, 180.0, znear=100.0, zfar=ZFAR)
    dev = torch.device("cuda:0")
    full = {"bgr": torch.empty((H, W, 3), dtype=torch.uint8, device=dev),
            "ranges": torch.empty((H, W), dtype=torch.float32, device=dev),
            "index": torch.empty((H, W), dtype=torch.int32, device=dev),
            "z24": torch.empty((H, W), dtype=torch.int32, device=dev)}
    h.render_device(full["bgr"].data_ptr(), full["ranges"].data_ptr(), full["index"].data_ptr(), full["z24"].data_ptr())
    h.sync()
    yield h, dems, full
    h.close()


def _render_sector(h, c0, c1, raster=0):
    import torch
    dev = torch.device("cuda:0")
    SW = c1 - c0
    out = {"bgr": torch.empty((H, SW, 3), dtype=torch.uint8, device=dev),
           "ranges": torch.empty((H, SW), dtype=torch.float32, device=dev),
           "index": torch.empty((H, SW), dtype=torch.int32, device=dev),
           "z24": torch.empty((H, SW), dtype=torch.int32, device=dev)}
    h.set_raster(raster)
    h.set_sector(c0, c1)
    try:
        h.render_device(out["bgr"].data_ptr(), out["ranges"].data_ptr(), out["index"].data_ptr(), out["z24"].data_ptr())
        h.sync()
    finally:
        h.set_sector(0, W)
        h.set_raster(0)
    return out


def test_the_render_is_not_trivial(cfg5):
    h, _, full = cfg5
    terrain = full["index"] != -1
    frac = float(terrain.float().mean())
    assert 0.2 < frac < 0.8
    ids = full["index"][terrain].cpu().numpy().view(np.uint32)
    assert int(ids.max()) < 2 * (2 * R - 1) ** 2
    assert bool(((full["ranges"] > 0) == terrain).all())


def test_one_sixteenth_sector_equals_the_oracle_on_every_output(cfg5):
    h, dems, full = cfg5
    c0, c1 = 5 * W // 16, 6 * W // 16
    m = h.mosaic()
    od = oracle.Dem(LAT, LON, dems, radius_cells=R, srtm1=True)
    assert od.N == m.shape[0] == 2 * R
    v = od.view(LAT, LON, W, H, -180.0, 180.0, znear=100.0, zfar=ZFAR)
    assert {k: np.float32(x) for k, x in v.as_dict().items()} == {k: np.float32(x) for k, x in h.view().items()}
    # rows of the device mosaic against the oracle's own reading of the tiles
    for j in (0, 1, 3599, 3600, 3601, 19799, 19800, 39599):
        assert np.array_equal(m[j, ::7], np.array([od.sample(i, j) for i in range(0, 2 * R, 7)], np.int16)), j
    orc = oracle.render(m, v, W, H, c0, c1)
    sect = _render_sector(h, c0, c1)
    for k in ("index", "z24", "bgr", "ranges"):
        got = sect[k].cpu().numpy()
        want = orc[k] if k != "z24" else orc[k].view(np.int32)
        assert np.array_equal(got, want), f"cfg5 sector [{c0},{c1}) vs oracle: {k}"
        assert np.array_equal(full[k][:, c0:c1].cpu().numpy(), want), f"cfg5 full render, columns [{c0},{c1}) vs oracle: {k}"


def test_triangle_ids_beyond_2_to_the_31_against_the_oracle(cfg5):
    """seen from the ground nothing north of the 27000th cell row is visible; from 9 km up the
    northern half of the window is, and with it triangle ids that need all 32 bits of the index
    map (3.136 G triangles): a narrow sector looking north, every output against the oracle"""
    h, dems, _ = cfg5
    c0, c1 = W // 2 - 256, W // 2 + 256
    od = oracle.Dem(LAT, LON, dems, radius_cells=R, srtm1=True)
    v = od.view(LAT, LON, W, H, -180.0, 180.0, viewer_z=9000.0, znear=100.0, zfar=ZFAR)
    m = h.mosaic()
    orc = oracle.render(m, v, W, H, c0, c1)
    ids = orc["index"][orc["index"] != -1].view(np.uint32)
    assert int(ids.max()) >= 2 ** 31 and int(ids.max()) < 2 * (2 * R - 1) ** 2
    hip = hzutil.hip_render(m, v, W, H, c0, c1)
    hzutil.assert_same_render(hip, orc, "cfg5 from 9 km up, looking north")


@pytest.mark.parametrize("raster", [0, 1])
def test_eight_sectors_tile_the_panorama(cfg5, raster):
    import torch
    h, _, full = cfg5
    for g in range(8):
        c0, c1 = g * W // 8, (g + 1) * W // 8
        if raster == 1 and g not in (0, 5):         # the first-design rasteriser is slow at this size: two sectors of it
            continue
        s = _render_sector(h, c0, c1, raster=raster)
        for k in s:
            assert torch.equal(s[k], full[k][:, c0:c1]), f"sector {g} raster {raster}: {k}"


def test_run_to_run(cfg5):
    import torch
    h, _, full = cfg5
    again = _render_sector(h, 0, W)
    for k in again:
        assert torch.equal(again[k], full[k]), k
