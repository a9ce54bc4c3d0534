"""Texture path ("next" row N4) on the GPU: the textured resolve against the oracle,
which is pinned to the reference's shaders on llvmpipe (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

import hzutil
import oracle

pytestmark = pytest.mark.gpu

LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON


def _texels(t, seed):
    """a map-like texture: smooth colour fields plus texel noise (the worst case for the sampler)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:t.tex_h, 0:t.tex_w]
    base = np.stack([127 + 120*np.sin(xx/37.0 + yy/91.0), 127 + 120*np.cos(xx/53.0), 127 + 120*np.sin(yy/29.0)], -1)
    return np.clip(base + rng.integers(-40, 41, base.shape), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("raster", [1, 2])
@pytest.mark.parametrize("case", ["small", "clipped", "partial", "moved"])
def test_textured_render_equals_oracle(case, raster):
    R, W, H, az0, az1, kw, lat, lon = {
        "small":   (64,  512,  128, -180, 180, dict(zfar=8000.0), LAT, LON),
        # big triangles next to the viewer cross the bottom of the image, the far sphere cuts the rest
        "clipped": (300, 1200, 300, -180, 180, dict(zfar=9000.0, viewer_z=2600.0), LAT, LON),
        "partial": (200, 900,  500, 20, 95,    dict(zfar=30000.0, znear_color=500.0, zfar_color=12000.0), LAT, LON),
        "moved":   (150, 1000, 250, -180, 180, dict(zfar=20000.0), LAT + 0.04, LON - 0.05),
    }[case]
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    v = od.view(lat, lon, W, H, az0, az1, **kw)
    t = od.texture(LAT, LON, viewer_lat=lat)
    texels = _texels(t, 3)
    orc = oracle.render(m, v, W, H, tex=t, texels=texels)
    hip = hzutil.hip_render(m, v, W, H, raster=raster, tex=t, texels=texels)
    hzutil.assert_same_render(hip, orc, f"textured {case}")
    plain = oracle.render(m, v, W, H, want=("bgr",))
    assert not np.array_equal(plain["bgr"], orc["bgr"])          # the texture really is in the picture
    assert (orc["bgr"][..., 1][orc["index"] >= 0] > 0).any()     # green only comes from the texture


def test_textured_sector_tiles_the_panorama():
    R, W, H = 128, 1001, 250
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    v = od.view(LAT, LON, W, H, -180, 180, zfar=15000.0)
    t = od.texture(LAT, LON)
    texels = _texels(t, 4)
    full = hzutil.hip_render(m, v, W, H, tex=t, texels=texels)
    for c0, c1 in [(0, 333), (333, 700), (700, 1001)]:
        part = hzutil.hip_render(m, v, W, H, col0=c0, col1=c1, tex=t, texels=texels)
        assert np.array_equal(part["bgr"], full["bgr"][:, c0:c1])


# ---- through the reference's API: tiles from disk, caller-supplied mosaic, moves ----

def _write_tiles(root, name, lowest_x, lowest_y, nx, ny, seed):
    """zoom-12 map tiles as PNG files in the layout the reference reads
    (dir_tiles/tiles_name/12/X/Y.png), in the PNG flavours tile servers emit; returns the
    mosaic as the reference builds it: uint8[ny*256, nx*256, 3] B,G,R, southern row first"""
    from PIL import Image
    rng = np.random.default_rng(seed)
    mosaic = np.zeros((ny * 256, nx * 256, 3), np.uint8)
    highest_y = lowest_y + ny - 1
    k = 0
    for ty in range(lowest_y, lowest_y + ny):
        for tx in range(lowest_x, lowest_x + nx):
            yy, xx = np.mgrid[0:256, 0:256]
            rgb = np.stack([(xx * 3 + tx * 40) % 256, (yy * 5 + ty * 17) % 256, (xx + yy + 31 * k) % 256], -1).astype(np.uint8)
            rgb = (rgb.astype(int) + rng.integers(-20, 21, rgb.shape)).clip(0, 255).astype(np.uint8)
            img = Image.fromarray(rgb, "RGB")
            flavour = k % 4
            if flavour == 1:            # palettised, as OSM's own tiles are (reference :339-352)
                img = img.quantize(colors=200)
                rgb = np.asarray(img.convert("RGB"))
            elif flavour == 2:          # with an alpha channel, which is dropped
                img = img.convert("RGBA")
            elif flavour == 3:          # grey
                img = img.convert("L")
                rgb = np.repeat(np.asarray(img)[..., None], 3, -1)
            d = root / name / "12" / str(tx)
            d.mkdir(parents=True, exist_ok=True)
            img.save(d / f"{ty}.png")
            x0, y0 = (tx - lowest_x) * 256, (highest_y - ty) * 256
            mosaic[y0:y0 + 256, x0:x0 + 256] = rgb[::-1, :, ::-1]      # bottom-up rows, B,G,R
            k += 1
    return mosaic


def test_render_texture_through_the_api(tmp_path):
    import horizonator_amd
    R, W, H = 64, 640, 160
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    t = od.texture(LAT, LON)
    texels = _write_tiles(tmp_path, "mapnik", t.lowest_x, t.lowest_y, t.ntiles_x, t.ntiles_y, 9)

    h = horizonator_amd.horizonator(LAT, LON, W, H, render_texture=True, dir_dems=d, dir_tiles=str(tmp_path),
                                    allow_downloads=False, render_radius_cells=R)
    try:
        assert h.texture_layout() == (t.lowest_x, t.lowest_y, t.ntiles_x, t.ntiles_y)
        image, ranges = h.render(-180, 180, zfar=9000.0)
        v = od.view(LAT, LON, W, H, -180, 180, zfar=9000.0)
        ref = oracle.render(m, v, W, H, tex=t, texels=texels)
        assert np.array_equal(image, ref["bgr"]) and np.array_equal(ranges, ref["ranges"])

        # moving the viewer moves the Taylor expansion of the tile projection (reference :707-759)
        lat, lon = LAT + 0.02, LON - 0.01
        image2, _ = h.render(-180, 180, lat=lat, lon=lon, zfar=9000.0)
        t2 = od.texture(LAT, LON, viewer_lat=lat)
        ref2 = oracle.render(m, od.view(lat, lon, W, H, -180, 180, zfar=9000.0), W, H, tex=t2, texels=texels)
        assert np.array_equal(image2, ref2["bgr"])

        # texturing off: the plain shaded image again; a caller-supplied mosaic: the same picture as from disk
        h.set_texture(None)
        plain, _ = h.render(-180, 180, lat=LAT, lon=LON, zfar=9000.0)
        assert np.array_equal(plain, oracle.render(m, v, W, H, want=("bgr",))["bgr"])
        h.set_texture(texels)
        again, _ = h.render(-180, 180, zfar=9000.0)
        assert np.array_equal(again, image)
        with pytest.raises(ValueError):
            h.set_texture(texels[:-1])
    finally:
        h.close()

    # a context created without a texture takes one later
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    try:
        h.set_texture(texels)
        image3, _ = h.render(-180, 180, zfar=9000.0)
        assert np.array_equal(image3, image)
    finally:
        h.close()

    # a tile that is not 256x256, and one that is no PNG at all: init fails with a message
    from PIL import Image
    bad = tmp_path / "mapnik" / "12" / str(t.lowest_x) / f"{t.lowest_y}.png"
    Image.new("RGB", (128, 128)).save(bad)
    with pytest.raises(RuntimeError):
        horizonator_amd.horizonator(LAT, LON, W, H, render_texture=True, dir_dems=d, dir_tiles=str(tmp_path),
                                    allow_downloads=False, render_radius_cells=R)
    bad.write_bytes(b"not a png")
    with pytest.raises(RuntimeError):
        horizonator_amd.horizonator(LAT, LON, W, H, render_texture=True, dir_dems=d, dir_tiles=str(tmp_path),
                                    allow_downloads=False, render_radius_cells=R)
