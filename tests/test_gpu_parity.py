"""Parity tests proper: the HIP path, called through the C-ABI, against the CPU
oracle (bit-exact) and against the reference's own llvmpipe renders (bands)."""
import ctypes as C
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import hzutil
import oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RENDERS = sorted(os.path.basename(p)[len("render_"):-4] for p in glob.glob(os.path.join(GOLD, "render_*.npz")))
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
RASTERS = [1, 2]           # HZ_RASTER_SCATTER, HZ_RASTER_MARCH


def _view(g):
    return oracle.make_view(**{k: float(g["u_" + k]) for k in oracle.VIEW_FIELDS})


@pytest.mark.parametrize("raster", RASTERS)
@pytest.mark.parametrize("name", RENDERS)
def test_golden_scenes_bit_exact_vs_oracle_and_vs_reference(name, raster):
    g = np.load(os.path.join(GOLD, f"render_{name}.npz"))
    W, H, v = int(g["W"]), int(g["H"]), _view(g)
    hip = hzutil.hip_render(g["mosaic"], v, W, H, raster=raster)
    orc = oracle.render(g["mosaic"], v, W, H)
    hzutil.assert_same_render(hip, orc, name)
    # and against what the reference's shaders drew on llvmpipe: identical
    assert np.array_equal(hip["bgr"], g["bgr"])
    assert np.array_equal(hip["z24"], g["z24"])


def _scene(R, W, H, az0, az1, lat=LAT, lon=LON, rough=False, **kw):
    d = hzutil.dem_dir_for(LAT, LON, R, rough=rough)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    return od.mosaic(), od.view(lat, lon, W, H, az0, az1, **kw)


SCENES = {
    "cfg1_zfar40k": dict(R=600, W=2000, H=500, az0=-180, az1=180),
    "cfg1_all_live": dict(R=600, W=2000, H=500, az0=-180, az1=180, zfar=200000.0),
    "rough_silhouettes": dict(R=300, W=1500, H=400, az0=-180, az1=180, zfar=60000.0, rough=True),
    "narrow_zoom": dict(R=300, W=1200, H=900, az0=40, az1=52, zfar=30000.0),
    "odd_sizes": dict(R=77, W=333, H=111, az0=-123.4, az1=77.7, zfar=9000.0),
    "high_viewer": dict(R=200, W=800, H=400, az0=-180, az1=180, viewer_z=6000.0, zfar=50000.0),
    "viewer_on_grid_vertex": dict(R=64, W=512, H=128, az0=-180, az1=180, lat=34.0 + 500 / 1200.0, lon=-118.0 + 500 / 1200.0),
    "near_clip_everything": dict(R=32, W=256, H=64, az0=-180, az1=180, znear=90000.0, zfar=100000.0),
    "colour_extents": dict(R=128, W=640, H=160, az0=-90, az1=90, znear=50.0, zfar=20000.0, znear_color=2000.0, zfar_color=3000.0),
}


@pytest.mark.parametrize("raster", RASTERS)
@pytest.mark.parametrize("name", sorted(SCENES))
def test_scenes_bit_exact_vs_oracle(name, raster):
    kw = dict(SCENES[name])
    R, W, H = kw.pop("R"), kw.pop("W"), kw.pop("H")
    mosaic, v = _scene(R, W, H, kw.pop("az0"), kw.pop("az1"), **kw)
    hip = hzutil.hip_render(mosaic, v, W, H, raster=raster)
    orc = oracle.render(mosaic, v, W, H)
    hzutil.assert_same_render(hip, orc, name)


@pytest.mark.parametrize("raster", RASTERS)
def test_sector_renders_tile_the_full_panorama(raster):
    """the azimuth-sector shard of each GPU is bit-identical to its columns of a full render"""
    mosaic, v = _scene(300, 1003, 250, -180, 180, zfar=60000.0)
    W, H = 1003, 250
    full = hzutil.hip_render(mosaic, v, W, H, raster=raster)
    from horizonator_amd.sharding import sector_columns
    for world in (2, 8):
        parts = [hzutil.hip_render(mosaic, v, W, H, *sector_columns(W, world, r), raster=raster) for r in range(world)]
        for k in full:
            assert np.array_equal(np.concatenate([p[k] for p in parts], axis=1), full[k]), (world, k)


def test_both_rasterisers_agree_and_are_deterministic_at_cfg2_size():
    """3x3-tile mosaic, 8000x2000 (BASELINE config 1): too big for the oracle in a
    test, so size-independent properties: run-to-run identical, two independent
    GPU rasterisers identical, sectors tile"""
    R, W, H = 1800, 8000, 2000
    mosaic, v = _scene(R, W, H, -180, 180, zfar=600000.0)
    a = hzutil.hip_render(mosaic, v, W, H, raster=1)
    b = hzutil.hip_render(mosaic, v, W, H, raster=2)
    hzutil.assert_same_render(a, b, "scatter vs columns")
    c = hzutil.hip_render(mosaic, v, W, H, raster=2)
    hzutil.assert_same_render(b, c, "run to run")
    s = hzutil.hip_render(mosaic, v, W, H, 3000, 4000, raster=2)
    for k in s:
        assert np.array_equal(s[k], b[k][:, 3000:4000]), k
    terrain = b["index"] >= 0
    assert 0.2 < terrain.mean() < 0.8
    # every visible triangle id is a real triangle, ranges are positive exactly on terrain
    assert b["index"].max() < 2 * (2 * R - 1) ** 2
    assert np.array_equal(b["ranges"] > 0, terrain)


def test_oracle_spot_check_at_cfg2_size():
    """one 1/16 azimuth sector of the cfg2 panorama against the oracle"""
    R, W, H = 1800, 8000, 2000
    mosaic, v = _scene(R, W, H, -180, 180, zfar=600000.0)
    hip = hzutil.hip_render(mosaic, v, W, H, 2500, 3000)
    orc = oracle.render(mosaic, v, W, H, 2500, 3000)
    hzutil.assert_same_render(hip, orc, "cfg2 sector")


@pytest.mark.parametrize("raster", RASTERS)
@pytest.mark.parametrize("capacity", [1, 50, 3000])
def test_full_triangle_queues_fall_back_correctly(raster, capacity, monkeypatch):
    """the HBM queues for medium/large triangles overflow: the kernels must
    rasterise the overflow in place and still match the oracle bit for bit"""
    monkeypatch.setenv("HZ_QUEUE_CAPACITY", str(capacity))
    mosaic, v = _scene(200, 1000, 250, -180, 180, zfar=30000.0)
    hip = hzutil.hip_render(mosaic, v, 1000, 250, raster=raster)
    sect = hzutil.hip_render(mosaic, v, 1000, 250, 300, 425, raster=raster)     # sector: medium queue in use
    orc = oracle.render(mosaic, v, 1000, 250)
    hzutil.assert_same_render(hip, orc, f"capacity {capacity}")
    for k in sect:
        assert np.array_equal(sect[k], orc[k][:, 300:425]), k


def test_full_medium_queue_on_a_reused_context(monkeypatch):
    """a context whose medium-triangle queue overflowed keeps records of earlier draws (other
    sectors, other viewpoints) in the slots the overflowing draw reserved but never wrote: the
    queue kernel must not read them.  Draws with different sectors and viewers on ONE context,
    each against the oracle."""
    monkeypatch.setenv("HZ_QUEUE_CAPACITY", "40")
    R, W, H = 200, 1000, 250
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    views = [(od.view(LAT, LON, W, H, -180, 180, zfar=30000.0), 600, 1000),
             (od.view(LAT + 0.01, LON - 0.02, W, H, -180, 180, zfar=30000.0), 0, 130),
             (od.view(LAT - 0.02, LON + 0.01, W, H, -100, 120, zfar=30000.0), 300, 425),
             (od.view(LAT, LON, W, H, -180, 180, zfar=30000.0), 0, 1000)]
    with hzutil.HipDev(m, W, H, raster=2) as dev:
        for rounds in range(2):
            for k, (v, c0, c1) in enumerate(views):
                got = dev.render(v, c0, c1)
                ref = oracle.render(m, v, W, H, c0, c1)
                hzutil.assert_same_render(got, ref, f"round {rounds} draw {k} sector [{c0},{c1})")


@pytest.mark.parametrize("clears", ["1", "0"])
def test_conversion_skips_only_what_nothing_was_drawn_into(clears, monkeypatch):
    """the conversion does not read 256-pixel row segments whose `touched` byte is zero.  One context,
    three framebuffers in turn: terrain that grows and shrinks between draws (viewer up, down, up), a
    sector of odd width in between (its conversion is the one-pixel-per-thread kernel, which clears
    words but leaves the bytes set), a draw that is never converted (cleared by memset before its
    framebuffer comes round again) - every conversion equal to the oracle's render of that view."""
    import ctypes as C
    monkeypatch.setenv("HZ_RESOLVE_CLEARS", clears)
    R, W, H = 200, 1500, 400
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    high = od.view(LAT, LON, W, H, -180, 180, viewer_z=6000.0, zfar=50000.0)
    low = od.view(LAT, LON, W, H, -180, 180, zfar=50000.0)
    zoom = od.view(LAT + 0.01, LON, W, H, 20, 75, zfar=30000.0)
    seq = [(high, 0, W), (low, 0, W), (zoom, 123, 770), (high, 0, W), (low, 256, 1280), (high, 0, W), (zoom, 0, W), (high, 0, W)]
    with hzutil.HipDev(m, W, H, raster=2) as dev:
        for k, (v, c0, c1) in enumerate(seq):
            if k == 4:
                # a draw nobody converts: its framebuffer stays dirty until the memset before its next turn
                vv = hzutil.hzlib.View()
                for name, _ in hzutil.hzlib.View._fields_:
                    setattr(vv, name, getattr(low, name))
                assert dev.lib.hz_hip_set_sector(dev.dev, 0, W) == 0
                assert dev.lib.hz_hip_draw(dev.dev, C.byref(vv)) == 0
            got = dev.render(v, c0, c1)
            ref = oracle.render(m, v, W, H, c0, c1)
            hzutil.assert_same_render(got, ref, f"draw {k} sector [{c0},{c1}) HZ_RESOLVE_CLEARS={clears}")


def test_two_round_draw_with_early_depth_test_changes_nothing(monkeypatch):
    """HZ_TWO_PASS=1: strips next to the viewer first, then everything else with the early
    depth test of mr_flush (hz_tri_depth_floor) - byte-identical to the one-round draw and to
    the oracle, on a scene where most of the far field is hidden and on one where none is"""
    R, W, H = 700, 4000, 1000
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    for viewer_z in (-1.0, 6000.0):
        v = od.view(LAT, LON, W, H, -180, 180, viewer_z=viewer_z, zfar=200000.0)
        monkeypatch.setenv("HZ_TWO_PASS", "0")
        one = hzutil.hip_render(m, v, W, H, raster=2)
        monkeypatch.setenv("HZ_TWO_PASS", "1")
        monkeypatch.setenv("HZ_NEAR_CELLS", "96")
        two = hzutil.hip_render(m, v, W, H, raster=2)
        hzutil.assert_same_render(two, one, f"two rounds vs one, viewer_z {viewer_z}")
        hzutil.assert_same_render(two, oracle.render(m, v, W, H), f"two rounds vs oracle, viewer_z {viewer_z}")


@pytest.mark.parametrize("capacity", [1, 70, 5000])
def test_two_round_draw_with_full_queues(capacity, monkeypatch):
    """the second round hands the few triangles that pass its early depth test to k_live through a
    queue of ids; with that queue (and all the others) too small the marching waves must draw the
    overflow themselves - same bytes"""
    monkeypatch.setenv("HZ_TWO_PASS", "1")
    monkeypatch.setenv("HZ_NEAR_CELLS", "64")
    monkeypatch.setenv("HZ_QUEUE_CAPACITY", str(capacity))
    R, W, H = 500, 3000, 750
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    v = od.view(LAT, LON, W, H, -180, 180, zfar=200000.0)
    hzutil.assert_same_render(hzutil.hip_render(m, v, W, H, raster=2), oracle.render(m, v, W, H), f"two rounds, capacity {capacity}")


def test_packed_strips_resolve_to_the_same_panorama():
    """the multi-GPU route on one GPU: every sector drawn and written as z24<<8 | red8 words
    (what a rank ships), then converted into the full-width outputs (what rank 0 does with the
    gathered strips) - the same bytes as the one-GPU render"""
    import torch
    import horizonator_amd
    R, W, H = 300, 1001, 250
    d = hzutil.dem_dir_for(LAT, LON, R)
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    try:
        image, ranges = h.render(-180, 180, zfar=30000.0)
        world = 3
        from horizonator_amd.sharding import sector_columns
        widest = -(-W // world)
        d_img = torch.full((H, W, 3), 77, dtype=torch.uint8, device="cuda:0")
        d_rng = torch.full((H, W), -7.0, dtype=torch.float32, device="cuda:0")
        strips = []
        for r in range(world):
            c0, c1 = sector_columns(W, world, r)
            h.set_sector(c0, c1)
            pk = torch.zeros((H, c1 - c0), dtype=torch.int32, device="cuda:0")
            h.render_packed(pk.data_ptr())
            h.sync()
            padded = torch.zeros((H, widest), dtype=torch.int32, device="cuda:0")    # as gather_strips pads them
            padded[:, :c1 - c0] = pk
            strips.append((padded, c0, c1 - c0))
        h.set_sector(0, W)
        for t, c0, n in strips:
            h.resolve_packed(t.data_ptr(), t.shape[1], n, c0, d_img.data_ptr(), d_rng.data_ptr())
        h.sync()
        assert np.array_equal(d_img.cpu().numpy(), image)
        assert np.array_equal(d_rng.cpu().numpy(), ranges)
        # ... and all strips in one call, as bench.py does with what the gather delivers
        d_img.fill_(0); d_rng.fill_(0)
        h.resolve_gathered(strips + [(strips[0][0], 0, 0)], d_img.data_ptr(), d_rng.data_ptr())
        h.sync()
        assert np.array_equal(d_img.cpu().numpy(), image) and np.array_equal(d_rng.cpu().numpy(), ranges)
        with pytest.raises(RuntimeError):
            h.resolve_packed(strips[0][0].data_ptr(), widest, widest, W - 5, d_img.data_ptr(), 0)
    finally:
        h.close()


def test_sparse_strips_resolve_to_the_same_panorama():
    """the default multi-GPU wire format on one GPU: every sector written as a sparse strip
    (terrain pixels only + mask + row bases), cut to header + terrain words as it would travel,
    then converted into the full-width outputs - the same bytes as the one-GPU render"""
    import torch
    import horizonator_amd
    from horizonator_amd.sharding import sector_columns, sparse_header_words, sparse_mask_stride
    R, W, H = 300, 1001, 250
    d = hzutil.dem_dir_for(LAT, LON, R)
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    try:
        image, ranges = h.render(-180, 180, zfar=30000.0)
        weights = [0.4, 1.0, 0.0, 1.3]
        world = len(weights)
        layout = [sector_columns(W, world, r, weights) for r in range(world)]
        widest = max(c1 - c0 for c0, c1 in layout)
        ms = sparse_mask_stride(widest)
        hdr = sparse_header_words(H, ms)
        strips, terrain = [], 0
        for c0, c1 in layout:
            buf = torch.full((hdr + H * widest,), -1, dtype=torch.int32, device="cuda:0")     # garbage beyond what is written
            if c1 > c0:
                h.set_sector(c0, c1)
                h.render_sparse(buf.data_ptr(), ms)
                h.sync()
                t = int(buf[0].item())
            else:
                buf.zero_()
                t = 0
            terrain += t
            sent = buf[:hdr + t].clone()                     # what travels
            strips.append((sent, c0, c1 - c0))
        assert terrain == int((ranges > 0).sum())            # one word per terrain pixel, none for the sky
        h.set_sector(0, W)
        d_img = torch.full((H, W, 3), 77, dtype=torch.uint8, device="cuda:0")
        d_rng = torch.full((H, W), -7.0, dtype=torch.float32, device="cuda:0")
        h.resolve_sparse_gathered([(t.data_ptr(), c0, n) for t, c0, n in strips], ms, d_img.data_ptr(), d_rng.data_ptr())
        h.sync()
        assert np.array_equal(d_img.cpu().numpy(), image)
        assert np.array_equal(d_rng.cpu().numpy(), ranges)
    finally:
        h.close()


RANDOM_GOLD = json.load(open(os.path.join(GOLD, "random_checksums.json")))


@pytest.mark.parametrize("seed", range(64))
def test_random_views_bit_exact_vs_oracle_and_vs_reference(seed):
    """seeded random viewpoints, azimuth extents (narrow, wide, wrapped, exactly 360), image
    sizes, depth/colour extents, viewer heights, sectors and rasterisers: every output equal to
    the oracle's, and image + depth equal (by hash) to what the reference's shaders drew on llvmpipe"""
    c = hzutil.random_view_case(seed)
    R, W, H = c["R"], c["W"], c["H"]
    d = hzutil.dem_dir_for(LAT, LON, R, rough=c["rough"])
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    v = od.view(c["lat"], c["lon"], W, H, c["az0"], c["az1"], **c["kw"])
    orc = oracle.render(m, v, W, H, c["c0"], c["c1"])
    for raster in RASTERS:
        hip = hzutil.hip_render(m, v, W, H, col0=c["c0"], col1=c["c1"], raster=raster)
        hzutil.assert_same_render(hip, orc, f"seed {seed}, raster {raster}: {c}")
    g = RANDOM_GOLD[str(seed)]
    # (the hashes pin the reference's render of ONE synthetic DEM: a generator that produced
    # other tiles here must not switch the comparison off silently)
    assert hashlib.sha256(m.tobytes()).hexdigest() == g["mosaic_sha256"], "the synthetic DEM differs on this machine"
    if True:
        full = hzutil.hip_render(m, v, W, H)
        assert hashlib.sha256(full["bgr"].tobytes()).hexdigest() == g["bgr_sha256"]
        assert hashlib.sha256(full["z24"].tobytes()).hexdigest() == g["z24_sha256"]


@pytest.mark.parametrize("capacity", [None, 3])
def test_tile_binned_rasteriser_draws_the_same_bytes(capacity, monkeypatch):
    """HZ_TILES=1 (hz_k_tile.h: BASELINE north_star's tile-binned rasteriser with per-bin depth in LDS, for the large
    triangles of every first round instead of those of zoomed views only): a whole image, a sector and a zoomed view against the oracle on every output; with tile lists
    of three triangles some tile's list overflows and the round must fall back to k_big - same bytes either way"""
    monkeypatch.setenv("HZ_TILES", "1")
    monkeypatch.setenv("HZ_TWO_PASS", "1")
    if capacity is not None:
        monkeypatch.setenv("HZ_TILE_LIST", str(capacity))
    R, W, H = 500, 3001, 750                                    # (odd width: partial tiles at the right edge)
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    for az0, az1, c0, c1 in ((-180, 180, 0, W), (-180, 180, 700, 1500), (10, 55, 0, W)):
        v = od.view(LAT, LON, W, H, az0, az1, zfar=200000.0)
        hzutil.assert_same_render(hzutil.hip_render(m, v, W, H, c0, c1, raster=2), oracle.render(m, v, W, H, c0, c1),
                                  f"tiles, az [{az0},{az1}], columns [{c0},{c1}), list of {capacity}")


@pytest.mark.parametrize("near_cells", [None, 24])
def test_coarse_depth_draws_the_same_bytes(near_cells, monkeypatch):
    """HZ_HIZ=1 (hz_k_hiz.h: the largest depth per 8x4 / 32x16 pixels, swept from the framebuffer after the first round
    and again before k_big; zoomed views have it on their own): whole images, sectors (tiles are counted from the
    sector's first column) and zoomed views of odd sizes (partial tiles at the right and bottom edges) against the
    oracle on every output.  With a first round of 24 cells the second round starts where cells are still tens of
    pixels wide: its boxes are the ones the tables are for."""
    monkeypatch.setenv("HZ_HIZ", "1")
    monkeypatch.setenv("HZ_TWO_PASS", "1")
    if near_cells is not None:
        monkeypatch.setenv("HZ_NEAR_CELLS", str(near_cells))
    R, W, H = 500, 3001, 749
    d = hzutil.dem_dir_for(LAT, LON, R, rough=near_cells is not None)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    for az0, az1, c0, c1 in ((-180, 180, 0, W), (-180, 180, 701, 1500), (10, 55, 0, W), (-8, 8, 0, W), (-8, 8, 1203, 2950)):
        v = od.view(LAT, LON, W, H, az0, az1, zfar=200000.0)
        hzutil.assert_same_render(hzutil.hip_render(m, v, W, H, c0, c1, raster=2), oracle.render(m, v, W, H, c0, c1),
                                  f"coarse depth, az [{az0},{az1}], columns [{c0},{c1}), first round of {near_cells} cells")
