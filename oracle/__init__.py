"""oracle - TEST INFRASTRUCTURE: ctypes binding of oracle/liboracle.so.

CPU restatement of the reference's DEM -> panorama path (see oracle/oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  Nothing under horizonator_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_DEM_PATH = os.path.join(_HERE, "_ref", "libdem_ref.so")
GLSL_GOLDEN_PATH = os.path.join(_HERE, "_ref", "glsl_golden")


class OrcDem(C.Structure):
    _fields_ = [
        ("cells_per_deg", C.c_int), ("radius_cells", C.c_int),
        ("origin_tile", C.c_int * 2), ("origin_cell", C.c_int * 2), ("ntiles", C.c_int * 2),
        ("tiles", C.c_void_p),
    ]


VIEW_FIELDS = ("viewer_cell_i", "viewer_cell_j", "viewer_z", "cos_viewer_lat", "deg_per_cell",
               "az_deg0", "az_deg1", "aspect", "znear", "zfar", "znear_color", "zfar_color")


class OrcView(C.Structure):
    _fields_ = [(n, C.c_float) for n in VIEW_FIELDS]

    def as_dict(self):
        return {n: getattr(self, n) for n in VIEW_FIELDS}


class OrcTex(C.Structure):
    _fields_ = [("viewer_lat_rad", C.c_float), ("origin_cell_lon_deg", C.c_float), ("origin_cell_lat_deg", C.c_float),
                ("lon0", C.c_float), ("lon1", C.c_float), ("dlat0", C.c_float), ("dlat1", C.c_float), ("dlat2", C.c_float),
                ("ntiles_x", C.c_int), ("ntiles_y", C.c_int), ("lowest_x", C.c_int), ("lowest_y", C.c_int),
                ("tex_w", C.c_int), ("tex_h", C.c_int), ("texels", C.c_void_p)]

    def as_glsl_job(self):
        """the dict oracle/glsl_run.py's texture functions take"""
        return dict(viewer_lat_rad=self.viewer_lat_rad, origin_cell_lon_deg=self.origin_cell_lon_deg,
                    origin_cell_lat_deg=self.origin_cell_lat_deg, texturemap_lon0=self.lon0, texturemap_lon1=self.lon1,
                    texturemap_dlat0=self.dlat0, texturemap_dlat1=self.dlat1, texturemap_dlat2=self.dlat2,
                    NtilesX=self.ntiles_x, NtilesY=self.ntiles_y, osmtile_lowestX=self.lowest_x,
                    osmtile_lowestY=self.lowest_y)


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.orc_dem_open.restype = C.c_int
    lib.orc_dem_open.argtypes = [P(OrcDem), C.c_float, C.c_float, C.c_int, C.c_float, C.c_char_p, C.c_int]
    lib.orc_dem_close.restype = None
    lib.orc_dem_close.argtypes = [P(OrcDem)]
    lib.orc_dem_sample.restype = C.c_int
    lib.orc_dem_sample.argtypes = [P(OrcDem), C.c_int, C.c_int]
    lib.orc_dem_mosaic.restype = None
    lib.orc_dem_mosaic.argtypes = [P(OrcDem), C.c_void_p]
    lib.orc_view_move.restype = None
    lib.orc_view_move.argtypes = [P(OrcView), P(OrcDem), C.c_float, C.c_float, C.c_float]
    lib.orc_vertex.restype = None
    lib.orc_vertex.argtypes = [P(OrcView), C.c_int, C.c_int, C.c_int, P(C.c_float * 4)]
    lib.orc_render.restype = C.c_int
    lib.orc_render.argtypes = [C.c_void_p, C.c_int, P(OrcView), C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    d = C.c_double
    lib.orc_project.restype = C.c_int
    lib.orc_project.argtypes = [P(d), P(d), P(d), d, d, d, d, d, d, d, d, d, C.c_int, C.c_int]
    lib.orc_unproject.restype = C.c_int
    lib.orc_unproject.argtypes = [P(C.c_float), P(C.c_float), C.c_int, C.c_int, d, d, d, d, d, d, d, C.c_int, C.c_int]
    lib.orc_link_cells.restype = C.c_int
    lib.orc_link_cells.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, d, d, d, d, C.c_void_p, C.c_void_p]
    lib.orc_poi_visibility.restype = None
    lib.orc_poi_visibility.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, d, d, d, d, d,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_osm_tile_id.restype = None
    lib.orc_osm_tile_id.argtypes = [P(C.c_int), P(C.c_int), C.c_float, C.c_float]
    lib.orc_tex_setup.restype = None
    lib.orc_tex_setup.argtypes = [P(OrcTex), P(OrcDem), C.c_float, C.c_float, C.c_float]
    lib.orc_vertex_tex.restype = None
    lib.orc_vertex_tex.argtypes = [P(OrcTex), C.c_float, C.c_int, C.c_int, P(C.c_float * 2)]
    lib.orc_tex_sample.restype = None
    lib.orc_tex_sample.argtypes = [P(OrcTex), C.c_float, C.c_float, P(C.c_uint8 * 3)]
    lib.orc_render_tex.restype = C.c_int
    lib.orc_render_tex.argtypes = [C.c_void_p, C.c_int, P(OrcView), P(OrcTex), C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_draw_triangles.restype = C.c_int
    lib.orc_draw_triangles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(OrcTex), C.c_void_p, C.c_void_p]
    lib.orc_tanel.restype = None
    lib.orc_tanel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
    _lib = lib
    return lib


class Dem:
    """the oracle's view of a DEM window (restates reference dem.c)"""

    def __init__(self, lat, lon, dir_dems, radius_cells=-1, radius_m=-1.0, srtm1=False):
        self.lib = load()
        self.d = OrcDem()
        rc = self.lib.orc_dem_open(C.byref(self.d), lat, lon, radius_cells, radius_m,
                                   str(dir_dems).encode(), int(srtm1))
        if rc != 0:
            raise RuntimeError(f"orc_dem_open failed ({rc})")

    @property
    def N(self):
        return 2 * self.d.radius_cells

    def sample(self, i, j):
        return self.lib.orc_dem_sample(C.byref(self.d), int(i), int(j))

    def mosaic(self):
        m = np.empty((self.N, self.N), np.int16)
        self.lib.orc_dem_mosaic(C.byref(self.d), m.ctypes.data)
        return m

    def view(self, lat, lon, W, H, az_deg0, az_deg1, viewer_z=-1.0,
             znear=100.0, zfar=40000.0, znear_color=-1.0, zfar_color=-1.0):
        """uniform values for a draw, derived as the reference's host code does"""
        v = OrcView()
        self.lib.orc_view_move(C.byref(v), C.byref(self.d), lat, lon, viewer_z)
        v.az_deg0, v.az_deg1 = az_deg0, az_deg1
        v.aspect = np.float32(W) / np.float32(H)          # reference horizonator-lib.c:658-659
        v.znear, v.zfar = znear, zfar
        v.znear_color = znear if znear_color < 0 else znear_color
        v.zfar_color = zfar if zfar_color < 0 else zfar_color
        return v

    def texture(self, init_lat, init_lon, viewer_lat=None):
        """uniforms and size of the texture mosaic for a context initialised at (init_lat, init_lon)
        whose viewer now stands at latitude viewer_lat (reference horizonator-lib.c:372-389,577-588,
        707-759,801-809); texels are attached by render()"""
        t = OrcTex()
        self.lib.orc_tex_setup(C.byref(t), C.byref(self.d), init_lat, init_lon,
                               init_lat if viewer_lat is None else viewer_lat)
        return t

    def close(self):
        if self.d.tiles:
            self.lib.orc_dem_close(C.byref(self.d))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_view(**kw):
    v = OrcView()
    for k, val in kw.items():
        setattr(v, k, val)
    return v


def render(mosaic, view, W, H, col0=0, col1=None, nthreads=0, want=("bgr", "ranges", "index", "z24"),
           tex=None, texels=None):
    """CPU render of image columns [col0,col1); returns a dict of arrays.
    tex (OrcTex) + texels uint8[tex_h,tex_w,3] (B,G,R; row 0 = south): the textured draw"""
    lib = load()
    mosaic = np.ascontiguousarray(mosaic, np.int16)
    N = mosaic.shape[0]
    if col1 is None:
        col1 = W
    SW = col1 - col0
    out = {}
    if "bgr" in want:
        out["bgr"] = np.empty((H, SW, 3), np.uint8)
    if "ranges" in want:
        out["ranges"] = np.empty((H, SW), np.float32)
    if "index" in want:
        out["index"] = np.empty((H, SW), np.int32)
    if "z24" in want:
        out["z24"] = np.empty((H, SW), np.uint32)

    def ptr(k):
        return out[k].ctypes.data if k in out else None

    if tex is not None:
        texels = np.ascontiguousarray(texels, np.uint8)
        if texels.shape != (tex.tex_h, tex.tex_w, 3):
            raise ValueError(f"texels must be uint8[{tex.tex_h},{tex.tex_w},3]")
        tex.texels = texels.ctypes.data
    rc = lib.orc_render_tex(mosaic.ctypes.data, N, C.byref(view), C.byref(tex) if tex is not None else None,
                            W, H, col0, col1,
                            ptr("bgr"), ptr("ranges"), ptr("index"), ptr("z24"), nthreads)
    if rc != 0:
        raise RuntimeError(f"orc_render failed ({rc})")
    return out


def draw_triangles(tris, W, H, texels=None):
    """raw clip-space triangles float32[n,3,6] (x,y,z, shade, s,t) through the oracle's clipper,
    rasteriser and, with texels uint8[th,tw,3] (B,G,R, row 0 = t 0), its textured fragment stage.
    Returns dict(bgr, z24), top row first like every other output."""
    lib = load()
    tris = np.ascontiguousarray(tris, np.float32)
    bgr = np.empty((H, W, 3), np.uint8)
    z24 = np.empty((H, W), np.uint32)
    tex = None
    if texels is not None:
        texels = np.ascontiguousarray(texels, np.uint8)
        tex = OrcTex()
        tex.tex_h, tex.tex_w = texels.shape[:2]
        tex.ntiles_x = tex.ntiles_y = 1
        tex.texels = texels.ctypes.data
    rc = lib.orc_draw_triangles(tris.ctypes.data, tris.shape[0], W, H, C.byref(tex) if tex is not None else None,
                                bgr.ctypes.data, z24.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_draw_triangles failed")
    return {"bgr": bgr[::-1].copy(), "z24": z24[::-1].copy()}


def vertices(mosaic, view):
    """gl_Position.xyz + red of every vertex: float32[N,N,4]"""
    lib = load()
    N = mosaic.shape[0]
    out = np.empty((N, N, 4), np.float32)
    buf = (C.c_float * 4)()
    for j in range(N):
        for i in range(N):
            lib.orc_vertex(C.byref(view), i, j, int(mosaic[j, i]), C.byref(buf))
            out[j, i] = buf[:]
    return out


def link_cells(ranges, cell_w, cell_h, cut_off_bottom_px, lat, lon, az_deg0, az_deg1):
    """reference annotator.c:228-264 on a range image; returns (lat, lon) float32[ny,nx]"""
    lib = load()
    ranges = np.ascontiguousarray(ranges, np.float32)
    H, W = ranges.shape
    nx = len(range(0, W - cell_w, cell_w))
    ny = len(range(0, H - cut_off_bottom_px - cell_h, cell_h))
    la = np.full((ny, nx), np.nan, np.float32)
    lo = np.full((ny, nx), np.nan, np.float32)
    if nx and ny:
        lib.orc_link_cells(ranges.ctypes.data, W, H, cut_off_bottom_px, cell_w, cell_h, lat, lon, az_deg0, az_deg1,
                           la.ctypes.data, lo.ctypes.data)
    return la, lo


def poi_visibility(ranges, pois, cut_off_bottom_px, lat, lon, az_deg0, az_deg1, ele_m):
    """reference annotator.c:280-348; pois float32[n,3] = lat, lon, ele_m"""
    lib = load()
    ranges = np.ascontiguousarray(ranges, np.float32)
    pois = np.ascontiguousarray(pois, np.float32)
    H, W = ranges.shape
    n = pois.shape[0]
    vis = np.zeros(n, np.uint8)
    x = np.zeros(n, np.float32)
    y = np.zeros(n, np.float32)
    lib.orc_poi_visibility(ranges.ctypes.data, W, H, cut_off_bottom_px, pois.ctypes.data, n, lat, lon, az_deg0, az_deg1,
                           ele_m, vis.ctypes.data, x.ctypes.data, y.ctypes.data)
    return vis, x, y


def tanel(W, H, az_deg0, az_deg1):
    lib = load()
    t = np.empty(H, np.float32)
    lib.orc_tanel(t.ctypes.data, W, H, az_deg0, az_deg1)
    return t


# ---- the reference's own dem.c, compiled in place (oracle/_ref) ---------------

class _RefDemCtx(C.Structure):
    # reference dem.h:10-29 (max_Ndems_ij = 4)
    _fields_ = [
        ("dems", (C.c_void_p * 4) * 4), ("mmap_sizes", (C.c_size_t * 4) * 4), ("mmap_fd", (C.c_int * 4) * 4),
        ("origin_dem_lon_lat", C.c_int * 2), ("origin_dem_cellij", C.c_int * 2), ("Ndems_ij", C.c_int * 2),
        ("radius_cells", C.c_int), ("cells_per_deg", C.c_int),
    ]


def load_ref_dem():
    """the reference's dem.c as oracle/_ref/libdem_ref.so, or None where absent"""
    if not os.path.exists(REF_DEM_PATH):
        return None
    lib = C.CDLL(REF_DEM_PATH)
    P = C.POINTER
    lib.horizonator_dem_init.restype = C.c_bool
    lib.horizonator_dem_init.argtypes = [P(_RefDemCtx), C.c_float, C.c_float, C.c_int, C.c_float, C.c_char_p, C.c_bool]
    lib.horizonator_dem_deinit.restype = None
    lib.horizonator_dem_deinit.argtypes = [P(_RefDemCtx)]
    lib.horizonator_dem_sample.restype = C.c_int16
    lib.horizonator_dem_sample.argtypes = [P(_RefDemCtx), C.c_int, C.c_int]
    lib.horizonator_dem_bounds_latlon_deg.restype = None
    lib.horizonator_dem_bounds_latlon_deg.argtypes = [P(_RefDemCtx)] + [P(C.c_float)] * 4
    return lib
