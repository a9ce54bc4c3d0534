"""A third, deliberately naive restatement of the reference's host arithmetic for the hot path - the uniforms of
horizonator_move / set_zextents (reference horizonator-lib.c:765-799, 864-885, with the window arithmetic of
dem.c:126-152 they build on) and the depth -> range loop of horizonator_render_offscreen (:1006-1047).

TEST CODE.  Written from the reference's text alone, statement by statement, in NumPy scalars; it shares no code
with oracle/ (the C restatement) nor with horizonator_amd/csrc/hz_host.c (the product), so that a mistake the two
have in common would show here (tests/test_naive_host_math.py).  The only things it borrows are glibc's own
cosf / tanf / hypotf through ctypes - the functions the reference itself calls - and DEM samples as data.

C semantics spelled out: `f32` values round after every operation; an expression with a double in it (M_PI, 1.0,
180.0 without the f) is evaluated in double from that operand on and rounded when it is assigned to a float.
"""
import ctypes
import ctypes.util

import numpy as np

f32, f64 = np.float32, np.float64
M_PI = f64(3.14159265358979323846)

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("cosf", "tanf", "floorf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.hypotf.restype = ctypes.c_float
_libm.hypotf.argtypes = [ctypes.c_float, ctypes.c_float]


def cosf(x):
    return f32(_libm.cosf(float(f32(x))))


def tanf(x):
    return f32(_libm.tanf(float(f32(x))))


def hypotf(a, b):
    return f32(_libm.hypotf(float(f32(a)), float(f32(b))))


def window(viewer_lat, viewer_lon, radius_cells, cells_per_deg=1200):
    """reference dem.c:139-152: origin tile (lon, lat) and origin cell (i, j) of the window"""
    tile, cell = [], []
    for v in (f32(viewer_lon), f32(viewer_lat)):
        # `viewer_lon_lat[i] * ctx->cells_per_deg`: float * int -> float; floor() takes it as a double
        icell_origin = int(np.floor(f64(v * f32(cells_per_deg)))) - (radius_cells - 1)
        origin_lon_lat = f32(icell_origin) / f32(cells_per_deg)
        t = int(np.floor(f64(origin_lon_lat)))
        # `(origin_lon_lat - origin_dem_lon_lat[i]) * cells_per_deg`: float arithmetic; round() = half away from zero
        x = f64((origin_lon_lat - f32(t)) * f32(cells_per_deg))
        c = int(np.sign(x) * np.floor(abs(x) + 0.5))
        tile.append(t)
        cell.append(c)
    return tile, cell


def move(viewer_lat, viewer_lon, origin_tile, origin_cell, sample, viewer_z=None, cells_per_deg=1200):
    """reference horizonator-lib.c:765-799 (+ :577 for deg_per_cell): the uniforms horizonator_move sets.
    sample(i, j) = horizonator_dem_sample; viewer_z None or < 0: stand on the terrain"""
    viewer_lat, viewer_lon = f32(viewer_lat), f32(viewer_lon)
    # (float - int) * int - int, all converted to float
    viewer_cell_i = (viewer_lon - f32(origin_tile[0])) * f32(cells_per_deg) - f32(origin_cell[0])
    viewer_cell_j = (viewer_lat - f32(origin_tile[1])) * f32(cells_per_deg) - f32(origin_cell[1])
    i0 = int(np.floor(viewer_cell_i))
    j0 = int(np.floor(viewer_cell_j))
    if viewer_z is None or viewer_z < 0:
        m = max(max(f32(sample(i0, j0)), f32(sample(i0 + 1, j0))), max(f32(sample(i0, j0 + 1)), f32(sample(i0 + 1, j0 + 1))))
        z = f32(f64(m) + f64(1.0))                          # `fmaxf(...) + 1.0`: a double addition, assigned to a float
    else:
        z = f32(viewer_z)
    # `cosf( viewer_lat * M_PI / 180.0f )`: double product, double quotient, converted to float for cosf
    cos_viewer_lat = cosf(f32(f64(viewer_lat) * M_PI / f64(f32(180.0))))
    return dict(viewer_cell_i=viewer_cell_i, viewer_cell_j=viewer_cell_j, viewer_z=z, cos_viewer_lat=cos_viewer_lat,
                deg_per_cell=f32(1.0) / f32(cells_per_deg))


def get_tanel(y, width, height, az_deg0, az_deg1):
    """reference horizonator-lib.c:1007-1012"""
    az_deg0, az_deg1 = f32(az_deg0), f32(az_deg1)
    aspect = f32(width) / f32(height)
    el_ndc = (f32(y) + f32(0.5)) / f32(height) * f32(2.0) - f32(1.0)
    # el_ndc * (az_deg1-az_deg0) / 2.f / aspect  in float, then * M_PI / 180.0f in double, assigned to a float
    el = f32(f64(el_ndc * (az_deg1 - az_deg0) / f32(2.0) / aspect) * M_PI / f64(f32(180.0)))
    return tanf(el)


def ranges_from_depth(z24, width, height, az_deg0, az_deg1, znear, zfar):
    """reference horizonator-lib.c:1013-1047 on the depth image as glReadPixels(GL_DEPTH_COMPONENT, GL_FLOAT) hands it
    out for a 24-bit depth buffer - float(z24 / (2^24 - 1)), pinned against llvmpipe by
    tests/test_oracle_golden.py::test_depth_readback_conversion -, including the vertical flip: z24[row] in GL row order
    (row 0 = bottom), result top row first.  Plain loops: small images only."""
    znear, zfar = f32(znear), f32(zfar)
    depth = (z24.astype(np.float64) / f64(16777215.0)).astype(np.float32)
    out = np.empty((height, width), np.float32)

    def rng(x, y, tanel):
        d = depth[y, x]
        if d == f32(1.0):
            return f32(-1.0)
        length_en = d * (zfar - znear) + znear
        z = f32(tanel) * length_en
        return hypotf(length_en, z)

    # the reference converts in place, swapping row y with row height-1-y as it goes
    res = np.empty((height, width), np.float32)
    for y in range(height // 2):
        tanel = get_tanel(y, width, height, az_deg0, az_deg1)
        for x in range(width):
            depth0 = rng(x, y, tanel)
            depth1 = rng(x, height - 1 - y, -tanel)
            res[y, x] = depth1
            res[height - 1 - y, x] = depth0
    if height & 1:
        y = height // 2
        tanel = get_tanel(y, width, height, az_deg0, az_deg1)
        for x in range(width):
            res[y, x] = rng(x, y, tanel)
    # `res` is the reference's buffer after its loop: row 0 of it is what was GL row height-1, i.e. the image's top row
    out[:] = res
    return out
