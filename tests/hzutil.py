"""shared helpers for the tests: synthetic DEM tiles, views, comparisons"""
import os
import tempfile

import numpy as np

from horizonator_amd import _lib as hzlib

# the survey's generic viewpoint: not on a grid sample
VIEW_LAT, VIEW_LON = 34.4137, -117.5621

_DEM_ROOT = os.environ.get("HZ_TEST_DEM_DIR", os.path.join(tempfile.gettempdir(), "hz_synth_dems"))


def dem_dir(lat_lo, lat_hi, lon_lo, lon_hi, srtm1=False, rough=False):
    """directory holding synthetic tiles covering the given integer lat/lon box
    (tiles are generated on first use and cached across test runs)"""
    name = ("srtm1" if srtm1 else "srtm3") + ("_rough" if rough else "")
    d = os.path.join(_DEM_ROOT, name)
    os.makedirs(d, exist_ok=True)
    gen = hzlib.load_demgen()
    rc = gen.hz_demgen_write_region(d.encode(), lat_lo, lat_hi, lon_lo, lon_hi, int(srtm1), int(rough))
    if rc < 0:
        raise RuntimeError(f"demgen failed ({rc})")
    return d


def tiles_for(lat, lon, radius_cells, srtm1=False):
    """integer lat/lon box a window of radius_cells around (lat,lon) can touch"""
    cpd = 3600 if srtm1 else 1200
    r = radius_cells / cpd + 0.01
    return (int(np.floor(lat - r)), int(np.floor(lat + r)),
            int(np.floor(lon - r)), int(np.floor(lon + r)))


def dem_dir_for(lat, lon, radius_cells, srtm1=False, rough=False):
    return dem_dir(*tiles_for(lat, lon, radius_cells, srtm1), srtm1=srtm1, rough=rough)


def viewpoint_lattice(lat=VIEW_LAT, lon=VIEW_LON, side=16, half_span_deg=0.2):
    """the batch of side*side viewpoints of BASELINE.json configs[3],
    a lattice of +-half_span_deg around (lat,lon); viewpoint v = row*side + column,
    rows south to north, columns west to east; float32 as the API takes them"""
    off = (np.arange(side, dtype=np.float64) / (side - 1) - 0.5) * 2.0 * half_span_deg
    lats = np.repeat(lat + off, side).astype(np.float32)
    lons = np.tile(lon + off, side).astype(np.float32)
    return lats, lons


def random_view_case(seed):
    """seeded random render configuration shared by the parity tests and by oracle/make_golden.py
    (tests/golden/random_checksums.json holds what the reference's shaders drew for each):
    viewpoint, azimuth extents (narrow, wide, wrapped, exactly 360), image size, depth/colour
    extents, viewer height, sector"""
    rng = np.random.default_rng(1000 + seed)
    c = {}
    c["R"] = int(rng.choice([24, 40, 75, 130, 200]))
    c["W"] = int(rng.integers(17, 900))
    c["H"] = int(rng.integers(9, 400))
    c["rough"] = bool(seed % 3 == 0)
    span = float(rng.choice([360.0, rng.uniform(0.5, 20.0), rng.uniform(20.0, 359.0)]))
    c["az0"] = float(rng.uniform(-720.0, 720.0))
    c["az1"] = c["az0"] + span
    frac = c["R"] / 1200.0 * 0.8
    c["lat"] = VIEW_LAT + float(rng.uniform(-frac, frac))
    c["lon"] = VIEW_LON + float(rng.uniform(-frac, frac))
    zfar = float(rng.choice([2000.0, 9000.0, 40000.0, 300000.0]))
    znear = float(rng.choice([100.0, 1.0, 500.0]))
    kw = dict(znear=znear, zfar=zfar)
    if seed % 4 == 1:
        kw.update(znear_color=float(rng.uniform(10.0, 3000.0)), zfar_color=float(rng.uniform(3500.0, 30000.0)))
    if seed % 5 == 2:
        kw.update(viewer_z=float(rng.uniform(0.0, 9000.0)))
    c["kw"] = kw
    c["c0"] = int(rng.integers(0, c["W"] - 1)) if seed % 2 else 0
    c["c1"] = int(rng.integers(c["c0"] + 1, c["W"] + 1)) if seed % 2 else c["W"]
    return c


def hash_texture(th, tw, seed=0, blocky=1):
    """a deterministic map-like texture without any RNG (integer hash of the texel index, so that
    fixtures only need to store its size and seed): uint8[th,tw,3], B,G,R, row 0 = southern edge.
    blocky > 1 repeats each value over blocky x blocky texels (flat areas with sharp borders)."""
    y, x = np.mgrid[0:th, 0:tw].astype(np.uint64)
    y //= np.uint64(blocky); x //= np.uint64(blocky)
    out = np.empty((th, tw, 3), np.uint8)
    for c in range(3):
        h = (x * np.uint64(73856093)) ^ (y * np.uint64(19349663)) ^ np.uint64((c + 1) * 83492791 + seed * 2654435761)
        h = (h ^ (h >> np.uint64(13))) * np.uint64(0x9E3779B97F4A7C15)
        out[..., c] = ((h >> np.uint64(40)) & np.uint64(255)).astype(np.uint8)
    return out


# ---- driving the HIP path through its C-ABI (include/hz_hip.h) ---------------

def hip_available():
    try:
        return hzlib.load().hz_hip_device_count() > 0
    except Exception:
        return False


class HipDev:
    """one device context of the C-ABI shim (include/hz_hip.h), reusable for several draws:
    hz_hip_create / upload_mosaic, then render(view, col0, col1) = set_sector / render_to_host
    (or draw / resolve_to_host) as often as wanted"""

    def __init__(self, mosaic, W, H, raster=0):
        self.lib = hzlib.load()
        mosaic = np.ascontiguousarray(mosaic, np.int16)
        self.W, self.H = W, H
        self.dev = self.lib.hz_hip_create(0, mosaic.shape[0], W, H)
        if not self.dev:
            raise RuntimeError("hz_hip_create failed: " + self.lib.hz_hip_last_error().decode())
        try:
            assert self.lib.hz_hip_upload_mosaic(self.dev, mosaic.ctypes.data) == 0
            assert self.lib.hz_hip_set_raster(self.dev, raster) == 0
        except Exception:
            self.close()
            raise

    def close(self):
        if self.dev:
            self.lib.hz_hip_destroy(self.dev)
            self.dev = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_texture(self, tex, texels):
        """texture path: `tex` is anything with the hz_texparams_t field names (e.g. oracle.OrcTex)"""
        import ctypes as C
        tp = hzlib.TexParams()
        for name, _ in hzlib.TexParams._fields_:
            setattr(tp, name, getattr(tex, name))
        texels = np.ascontiguousarray(texels, np.uint8)
        assert texels.shape == (tp.tex_h, tp.tex_w, 3)
        assert self.lib.hz_hip_set_texture(self.dev, C.byref(tp), texels.ctypes.data) == 0, self.lib.hz_hip_last_error()

    def render(self, view, col0=0, col1=None, tanel=None):
        """`view` is anything with the hz_view_t field names as attributes (e.g. oracle.OrcView)"""
        import ctypes as C
        lib, W, H = self.lib, self.W, self.H
        if col1 is None:
            col1 = W
        SW = col1 - col0
        assert lib.hz_hip_set_sector(self.dev, col0, col1) == 0
        v = hzlib.View()
        for name, _ in hzlib.View._fields_:
            setattr(v, name, getattr(view, name))
        if tanel is None:
            # tan(elevation) per GL row exactly as hz_host.c / the reference derive it
            import oracle
            tanel = oracle.tanel(W, H, v.az_deg0, v.az_deg1)
        tanel = np.ascontiguousarray(tanel, np.float32)
        out = {"bgr": np.empty((H, SW, 3), np.uint8), "ranges": np.empty((H, SW), np.float32),
               "index": np.empty((H, SW), np.int32), "z24": np.empty((H, SW), np.uint32)}
        # draw + delivery into host memory as one call (hz_hostpath.cpp: in azimuth sectors where the image is large or the
        # context's host_sectors option says so); every other call the two-step form, hz_hip_draw then hz_hip_resolve_to_host
        self.calls = getattr(self, "calls", 0) + 1
        if self.calls % 2:
            rc = lib.hz_hip_render_to_host(self.dev, C.byref(v), tanel.ctypes.data, out["bgr"].ctypes.data,
                                           out["ranges"].ctypes.data, out["index"].ctypes.data, out["z24"].ctypes.data)
        else:
            assert lib.hz_hip_draw(self.dev, C.byref(v)) == 0, lib.hz_hip_last_error()
            rc = lib.hz_hip_resolve_to_host(self.dev, C.byref(v), tanel.ctypes.data, out["bgr"].ctypes.data,
                                            out["ranges"].ctypes.data, out["index"].ctypes.data, out["z24"].ctypes.data)
        assert rc == 0, lib.hz_hip_last_error()
        return out


def hip_render(mosaic, view, W, H, col0=0, col1=None, raster=0, tanel=None, tex=None, texels=None):
    """mosaic int16[N,N] + uniform values -> dict(bgr, ranges, index, z24) via
    hz_hip_create / upload_mosaic / draw / resolve_to_host on a fresh context"""
    with HipDev(mosaic, W, H, raster=raster) as dev:
        if tex is not None:
            dev.set_texture(tex, texels)
        return dev.render(view, col0, col1, tanel=tanel)


def assert_same_render(a, b, what=""):
    for k in ("index", "z24", "bgr", "ranges"):
        if k in a and k in b:
            if not np.array_equal(a[k], b[k]):
                bad = np.argwhere(a[k] != b[k])
                raise AssertionError(f"{what}: {k} differs at {len(bad)} places, first {bad[0]}: "
                                     f"{a[k][tuple(bad[0])]} vs {b[k][tuple(bad[0])]}")
