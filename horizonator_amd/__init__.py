"""horizonator_amd - SRTM terrain panoramas rendered by hand-written HIP kernels on MI355X.

Host-side mirror of the reference's Python surface (reference
horizonator-pywrap.c): the same type name, constructor arguments, `render`
keywords, defaults, return shapes and error behaviour, on top of the C-ABI of
libhorizonator.so.  The reference's module is a CPython extension written in C;
its toolchain pieces (generated docstring headers, mrbuild) are not in this
image, so the wrapper is restated here over ctypes.  INTEGRATION.md shows how
the reference's own horizonator-pywrap.c is built against this library instead.

    h = horizonator_amd.horizonator(34.4137, -117.5621, 2000, 500, dir_dems=...,
                                    render_radius_cells=600)
    image, ranges = h.render(-180, 180)      # uint8[H,W,3] BGR, float32[H,W]
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RASTER_AUTO, RASTER_MARCH, RASTER_SCATTER, Options, Times, View  # noqa: F401

HORIZONATOR_ZNEAR_DEFAULT = 100.0
HORIZONATOR_ZFAR_DEFAULT = 40000.0

__all__ = ["horizonator", "RASTER_AUTO", "RASTER_SCATTER", "RASTER_MARCH"]


def _enc(s):
    return None if s is None else str(s).encode()


class _ResultMemory:
    """The arrays render() returns, recycled.  The reference's wrapper makes new arrays for every call
    (horizonator-pywrap.c:234-250, PyArray_SimpleNew): 448 MB of pages the kernel has to map and zero while the call
    writes them - 2 ms of a 5 ms call for a 16000x4000 panorama, where the render and its transfer take 3.  A caller
    that loops `image, ranges = h.render(...)` drops the previous results as it goes: their memory, pages mapped and
    on the NUMA node the library's threads first wrote them from, is what the next call's results are made of.
    Memory still referenced - by the caller's arrays or any view of them - is never handed out again (its reference
    count says so); at most two calls' worth of dropped memory is kept.  HZ_PY_RECYCLE=0: plain np.empty()."""

    def __init__(self):
        import os
        self._bufs = []                 # uint8 owners, the most recently handed out last
        import sys
        self._on = os.environ.get("HZ_PY_RECYCLE", "1") != "0" and hasattr(sys, "getrefcount")

    def take(self, specs):
        """arrays of the given (shape, dtype) list, contents undefined"""
        import sys
        if not self._on:
            return [np.empty(shape, dt) for shape, dt in specs]
        out = []
        for shape, dt in specs:
            n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dt).itemsize
            b = None
            for k in range(len(self._bufs)):
                # (the list's reference and getrefcount's own argument: nobody else holds the owner or a view of it)
                if self._bufs[k].nbytes == n and sys.getrefcount(self._bufs[k]) == 2:
                    b = self._bufs.pop(k)
                    break
            if b is None:
                b = np.empty(n, np.uint8)
            self._bufs.append(b)
            out.append(b.view(dt).reshape(shape))
            del b
        # two calls' worth at most: the oldest go (memory the caller still holds lives on through the caller's arrays)
        del self._bufs[:max(0, len(self._bufs) - 2 * len(specs))]
        return out

    def clear(self):
        self._bufs = []


class horizonator:
    """SRTM terrain renderer (reference horizonator.docstring, horizonator-pywrap.c:49-125).

    horizonator(lat, lon, width, height, render_texture=False, SRTM1=False,
                dir_dems=None, dir_tiles=None, tiles_name=None, tiles_url_fmt=None,
                allow_downloads=True, render_radius_cells=-1, render_radius_m=-1.)

    The constructor loads the DEM window around (lat, lon) into HBM (slow);
    render() calls are fast and may move the viewer inside that window.
    """

    # reference horizonator-pywrap.c:65
    _render_radius_cells_default = 1000

    def __init__(self, lat, lon, width, height,
                 render_texture=False, SRTM1=False,
                 dir_dems=None, dir_tiles=None,
                 tiles_name=None, tiles_url_fmt=None,
                 allow_downloads=True,
                 render_radius_cells=-1, render_radius_m=-1.0):
        self._lib = _lib.load()
        if getattr(self, "_ctx", None) is not None and self._ctx.offscreen.inited:
            # reference horizonator-pywrap.c:81-85
            raise RuntimeError("Trying to init an already-inited object")
        self._ctx = _lib.Context()
        width, height = int(width), int(height)
        if width < 0 or height < 0:
            raise OverflowError("can't convert negative value to unsigned int")
        render_radius_cells = int(render_radius_cells)
        render_radius_m = float(render_radius_m)
        # reference horizonator-pywrap.c:98-104
        if render_radius_cells < 0 and render_radius_m < 0:
            render_radius_cells = self._render_radius_cells_default
        elif render_radius_cells > 0 and render_radius_m > 0:
            raise RuntimeError("both render_radius_cells,render_radius_m cannot be >0")

        ok = self._lib.horizonator_init(
            C.byref(self._ctx), float(lat), float(lon), None,
            width, height, render_radius_cells, render_radius_m,
            True, bool(render_texture), bool(SRTM1),
            _enc(dir_dems), _enc(dir_tiles), _enc(tiles_name), _enc(tiles_url_fmt),
            bool(allow_downloads))
        if not ok:
            self._ctx = None
            raise RuntimeError("horizonator_init() failed")

    @classmethod
    def from_mosaic(cls, lat, lon, width, height, window, mosaic):
        """A context over a DEM window that another process loaded (multi-GPU: rank 0 reads the
        tiles, the others receive `window` = (cells_per_deg, radius_cells, origin_tile_lon,
        origin_tile_lat, origin_cell_i, origin_cell_j) and the int16 mosaic; see
        sharding.broadcast_dem).  include/horizonator_amd.h: horizonator_amd_init_from_mosaic."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._ctx = _lib.Context()
        w = _lib.Window()
        w.cells_per_deg, w.radius_cells = int(window[0]), int(window[1])
        w.origin_tile[0], w.origin_tile[1] = int(window[2]), int(window[3])
        w.origin_cell[0], w.origin_cell[1] = int(window[4]), int(window[5])
        mosaic = np.ascontiguousarray(mosaic, np.int16)
        n = 2 * w.radius_cells
        if mosaic.shape != (n, n):
            raise ValueError(f"the mosaic of this window is int16[{n},{n}]")
        ok = self._lib.horizonator_amd_init_from_mosaic(C.byref(self._ctx), float(lat), float(lon), None,
                                                        int(width), int(height), C.byref(w), mosaic.ctypes.data)
        if not ok:
            self._ctx = None
            raise RuntimeError("horizonator_amd_init_from_mosaic() failed")
        return self

    def window(self):
        """(cells_per_deg, radius_cells, origin_tile_lon, origin_tile_lat, origin_cell_i, origin_cell_j)"""
        w = _lib.Window()
        if not self._lib.horizonator_amd_get_window(C.byref(self._ctx), C.byref(w)):
            raise RuntimeError("horizonator_amd_get_window() failed")
        return (w.cells_per_deg, w.radius_cells, w.origin_tile[0], w.origin_tile[1], w.origin_cell[0], w.origin_cell[1])

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None:
            self._lib.horizonator_deinit(C.byref(self._ctx))
            self._ctx = None
        if getattr(self, "_results", None) is not None:
            self._results.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __str__(self):
        # reference horizonator-pywrap.c:133-156: at most 9 characters of each
        lat = repr(float(self._ctx.viewer_lat))[:9]
        lon = repr(float(self._ctx.viewer_lon))[:9]
        return f"Looking out from {lat},{lon}"

    # -- properties of the loaded window -----------------------------------
    @property
    def width(self):
        return self._ctx.offscreen.width

    @property
    def height(self):
        return self._ctx.offscreen.height

    @property
    def radius_cells(self):
        return self._ctx.dems.radius_cells

    @property
    def Ntriangles(self):
        return self._ctx.Ntriangles

    @property
    def sector(self):
        return getattr(self, "_sector", (0, self.width))

    def _result_memory(self):
        m = getattr(self, "_results", None)
        if m is None:
            m = self._results = _ResultMemory()
        return m

    # -- the reference's render() -------------------------------------------
    def _prepare(self, az_deg0, az_deg1, lat, lon, az_extents_use_pixel_centers,
                 znear, zfar, znear_color, zfar_color):
        az_deg0, az_deg1 = float(az_deg0), float(az_deg1)
        # reference horizonator-pywrap.c:194-195
        if znear_color < 0.0:
            znear_color = znear
        if zfar_color < 0.0:
            zfar_color = zfar
        if az_extents_use_pixel_centers:
            # reference horizonator-pywrap.c:204-212: the caller's azimuths are
            # those of the first and last pixel CENTRES; widen by half a pixel
            az_per_pixel = (az_deg1 - az_deg0) / float(self._ctx.offscreen.width - 1)
            az_deg0 -= az_per_pixel / 2.0
            az_deg1 += az_per_pixel / 2.0
        ctx = C.byref(self._ctx)
        if not self._lib.horizonator_pan_zoom(ctx, az_deg0, az_deg1):
            raise RuntimeError("horizonator_pan_zoom() failed")
        if lat > -1000.0:
            if not self._lib.horizonator_move(ctx, None, float(lat), float(lon)):
                raise RuntimeError("horizonator_move() failed")
        if not self._lib.horizonator_set_zextents(ctx, znear, zfar, znear_color, zfar_color):
            raise RuntimeError("horizonator_set_zextents() failed")

    def render(self, az_deg0, az_deg1, lat=-1000.0, lon=-1000.0,
               return_image=True, return_range=True,
               az_extents_use_pixel_centers=False,
               znear=HORIZONATOR_ZNEAR_DEFAULT, zfar=HORIZONATOR_ZFAR_DEFAULT,
               znear_color=-1.0, zfar_color=-1.0):
        """render(az_deg0, az_deg1, lat=, lon=, return_image=True, return_range=True,
                  az_extents_use_pixel_centers=False, znear=100, zfar=40000,
                  znear_color=-1, zfar_color=-1)

        reference render.docstring / horizonator-pywrap.c:158-279.  Returns
        (image, ranges), or only one of them, or () if neither is asked for.
        image: uint8[H,W,3] BGR; ranges: float32[H,W], < 0 where no terrain.
        """
        if not return_image and not return_range:
            return ()                       # reference horizonator-pywrap.c:198-202
        self._prepare(az_deg0, az_deg1, lat, lon, az_extents_use_pixel_centers,
                      float(znear), float(zfar), float(znear_color), float(zfar_color))
        c0, c1 = self.sector
        H, W = self._ctx.offscreen.height, c1 - c0
        specs = ([((H, W, 3), np.uint8)] if return_image else []) + ([((H, W), np.float32)] if return_range else [])
        got = self._result_memory().take(specs)
        image = got.pop(0) if return_image else None
        ranges = got.pop(0) if return_range else None
        ok = self._lib.horizonator_render_offscreen(
            C.byref(self._ctx),
            image.ctypes.data if image is not None else None,
            ranges.ctypes.data if ranges is not None else None)
        if not ok:
            raise RuntimeError("horizonator_render_offscreen() failed")
        if return_image and not return_range:
            return image
        if return_range and not return_image:
            return ranges
        return image, ranges

    def render_into(self, image=None, ranges=None):
        """horizonator_render_offscreen() (reference horizonator.h:165-169) with the current view
        into caller-owned numpy arrays (uint8[H,W,3] / float32[H,W], C-contiguous; either may be
        None): what a C caller that keeps its buffers does."""
        c0, c1 = self.sector
        H, W = self._ctx.offscreen.height, c1 - c0
        for a, shape, dt in ((image, (H, W, 3), np.uint8), (ranges, (H, W), np.float32)):
            if a is not None and (a.shape != shape or a.dtype != dt or not a.flags.c_contiguous):
                raise ValueError("render_into() wants C-contiguous %s arrays of shape %s" % (np.dtype(dt).name, shape))
        if not self._lib.horizonator_render_offscreen(C.byref(self._ctx),
                                                      image.ctypes.data if image is not None else None,
                                                      ranges.ctypes.data if ranges is not None else None):
            raise RuntimeError("horizonator_render_offscreen() failed")

    # -- build-side additions ------------------------------------------------
    def render_begin(self, image=None, ranges=None):
        """the first half of render_into(): queues the draw of the current view into the caller's arrays and returns;
        render_end() returns when they hold the panorama.  Two panoramas may be in flight (each with arrays of its own):
        the device draws one while the other crosses PCIe.  The arrays must stay alive and untouched until the matching end."""
        c0, c1 = self.sector
        H, W = self._ctx.offscreen.height, c1 - c0
        for a, shape, dt in ((image, (H, W, 3), np.uint8), (ranges, (H, W), np.float32)):
            if a is not None and (a.shape != shape or a.dtype != dt or not a.flags.c_contiguous):
                raise ValueError("render_begin() wants C-contiguous %s arrays of shape %s" % (np.dtype(dt).name, shape))
        if not self._lib.horizonator_amd_render_begin(C.byref(self._ctx), image.ctypes.data if image is not None else None,
                                                      ranges.ctypes.data if ranges is not None else None):
            raise RuntimeError("horizonator_amd_render_begin() failed")

    def render_end(self):
        if not self._lib.horizonator_amd_render_end(C.byref(self._ctx)):
            raise RuntimeError("horizonator_amd_render_end() failed")

    def render_full(self, az_deg0, az_deg1, lat=-1000.0, lon=-1000.0,
                    az_extents_use_pixel_centers=False,
                    znear=HORIZONATOR_ZNEAR_DEFAULT, zfar=HORIZONATOR_ZFAR_DEFAULT,
                    znear_color=-1.0, zfar_color=-1.0):
        """Like render() but also returns the visible-triangle index map and the
        raw 24-bit depth: (image, ranges, index int32[H,W], z24 uint32[H,W])."""
        self._prepare(az_deg0, az_deg1, lat, lon, az_extents_use_pixel_centers,
                      float(znear), float(zfar), float(znear_color), float(zfar_color))
        c0, c1 = self.sector
        H, W = self._ctx.offscreen.height, c1 - c0
        image, ranges, index, z24 = self._result_memory().take(
            [((H, W, 3), np.uint8), ((H, W), np.float32), ((H, W), np.int32), ((H, W), np.uint32)])
        ok = self._lib.horizonator_amd_render(
            C.byref(self._ctx), image.ctypes.data, ranges.ctypes.data,
            index.ctypes.data, z24.ctypes.data)
        if not ok:
            raise RuntimeError("horizonator_amd_render() failed")
        return image, ranges, index, z24

    def render_device(self, d_image=0, d_ranges=0, d_index=0, d_z24=0):
        """Draw with the current view into caller-owned DEVICE buffers (raw
        pointers, 0 = skip).  Asynchronous; call sync()."""
        ok = self._lib.horizonator_amd_render_device(
            C.byref(self._ctx), d_image or None, d_ranges or None, d_index or None, d_z24 or None)
        if not ok:
            raise RuntimeError("horizonator_amd_render_device() failed")

    def render_batch(self, lats, lons, d_images=0, d_ranges=0, viewer_z=None):
        """One render per viewpoint (lats[v], lons[v]) with the current azimuth
        and z extents, into caller-owned DEVICE buffers laid out [n][H][SW][3]
        (BGR) and [n][H][SW] (float32); raw pointers, 0 = skip.  The whole batch
        is queued without waiting; call sync().  Returns the viewer heights used
        (float32[n]): viewer_z, or 1 m above the terrain where that is None/<0."""
        lats = np.ascontiguousarray(lats, np.float32)
        lons = np.ascontiguousarray(lons, np.float32)
        if lats.shape != lons.shape or lats.ndim != 1:
            raise ValueError("lats and lons must be 1-D and of equal length")
        z = np.full(lats.shape, -1.0, np.float32) if viewer_z is None else \
            np.array(np.broadcast_to(np.asarray(viewer_z, np.float32), lats.shape))
        ok = self._lib.horizonator_amd_render_batch(
            C.byref(self._ctx), int(lats.size), lats.ctypes.data, lons.ctypes.data, z.ctypes.data,
            d_images or None, d_ranges or None)
        if not ok:
            raise RuntimeError("horizonator_amd_render_batch() failed")
        return z

    def render_packed(self, d_packed):
        """Draw with the current view and write this context's sector as uint32 z24<<8 | red8
        per pixel into the DEVICE buffer d_packed ([H, sector width], raw pointer): the 4-byte
        form in which strips travel between GPUs.  Asynchronous; call sync()."""
        if not self._lib.horizonator_amd_render_packed(C.byref(self._ctx), d_packed):
            raise RuntimeError("horizonator_amd_render_packed() failed")

    def resolve_packed(self, d_packed, packed_stride, ncols, out_col0, d_image=0, d_ranges=0):
        """Convert packed words (from any GPU, drawn with this context's view) into columns
        [out_col0, out_col0+ncols) of the full-width DEVICE outputs image uint8[H,W,3] and
        ranges float32[H,W] (raw pointers, 0 = skip).  Asynchronous."""
        if not self._lib.horizonator_amd_resolve_packed(C.byref(self._ctx), d_packed, int(packed_stride), int(ncols),
                                                        int(out_col0), d_image or None, d_ranges or None):
            raise RuntimeError("horizonator_amd_resolve_packed() failed")

    def resolve_gathered(self, parts, d_image=0, d_ranges=0):
        """parts as PendingGather.parts() returns them - [(tensor [H, stride], col0, ncols), ...] -
        converted into the full-width DEVICE outputs in one call"""
        n = len(parts)
        if n == 0:
            return
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t, _, _ in parts])
        ncols = (C.c_int * n)(*[int(k) for _, _, k in parts])
        col0 = (C.c_int * n)(*[int(c) for _, c, _ in parts])
        stride = int(parts[0][0].shape[1])
        if any(int(t.shape[1]) != stride for t, _, _ in parts):
            raise ValueError("gathered strips share one row stride")
        if not self._lib.horizonator_amd_resolve_packed_strips(C.byref(self._ctx), n, ptrs, stride, ncols, col0,
                                                               d_image or None, d_ranges or None):
            raise RuntimeError("horizonator_amd_resolve_packed_strips() failed")

    def render_sparse(self, d_out, mask_stride):
        """Draw and write this context's sector as a sparse strip (include/hz_hip.h: terrain
        pixels only, plus a mask) into the DEVICE buffer d_out (raw pointer, room for
        sharding.sparse_header_words(H, mask_stride) + H*sector_width uint32).  Asynchronous."""
        if not self._lib.horizonator_amd_render_sparse(C.byref(self._ctx), d_out, int(mask_stride)):
            raise RuntimeError("horizonator_amd_render_sparse() failed")

    def resolve_sparse_gathered(self, strips, mask_stride, d_image=0, d_ranges=0):
        """strips = [(device pointer of a sparse strip, col0, ncols), ...] converted into the
        full-width DEVICE outputs in one call"""
        n = len(strips)
        if n == 0:
            return
        ptrs = (C.c_void_p * n)(*[int(p) for p, _, _ in strips])
        ncols = (C.c_int * n)(*[int(k) for _, _, k in strips])
        col0 = (C.c_int * n)(*[int(c) for _, c, _ in strips])
        if not self._lib.horizonator_amd_resolve_sparse_strips(C.byref(self._ctx), n, ptrs, int(mask_stride), ncols, col0,
                                                               d_image or None, d_ranges or None):
            raise RuntimeError("horizonator_amd_resolve_sparse_strips() failed")

    def texture_layout(self):
        """(lowest_x, lowest_y, ntiles_x, ntiles_y): the zoom-12 slippy-map tiles the texture
        of this context is made of (reference horizonator-lib.c:372-389); the texture is
        ntiles_y*256 rows of ntiles_x*256 texels"""
        v = [C.c_int() for _ in range(4)]
        if not self._lib.horizonator_amd_texture_layout(C.byref(self._ctx), *[C.byref(x) for x in v]):
            raise RuntimeError("horizonator_amd_texture_layout() failed")
        return tuple(x.value for x in v)

    def set_texture(self, texels_bgr):
        """Drape a caller-supplied map over the terrain instead of tiles read from disk:
        uint8[ntiles_y*256, ntiles_x*256, 3], B,G,R, row 0 = southern edge.  None switches
        texturing off.  (include/horizonator_amd.h: horizonator_amd_set_texture)"""
        if texels_bgr is None:
            ok = self._lib.horizonator_amd_set_texture(C.byref(self._ctx), None)
        else:
            _, _, nx, ny = self.texture_layout()
            texels_bgr = np.ascontiguousarray(texels_bgr, np.uint8)
            if texels_bgr.shape != (ny * 256, nx * 256, 3):
                raise ValueError(f"the texture of this context is uint8[{ny * 256},{nx * 256},3]")
            ok = self._lib.horizonator_amd_set_texture(C.byref(self._ctx), texels_bgr.ctypes.data)
        if not ok:
            raise RuntimeError("horizonator_amd_set_texture() failed")

    def set_view(self, az_deg0, az_deg1, lat=-1000.0, lon=-1000.0,
                 znear=HORIZONATOR_ZNEAR_DEFAULT, zfar=HORIZONATOR_ZFAR_DEFAULT,
                 znear_color=-1.0, zfar_color=-1.0):
        self._prepare(az_deg0, az_deg1, lat, lon, False,
                      float(znear), float(zfar), float(znear_color), float(zfar_color))

    def sync(self):
        if not self._lib.horizonator_amd_sync(C.byref(self._ctx)):
            raise RuntimeError("horizonator_amd_sync() failed")

    def stream_waits_for_outputs(self, stream):
        """work queued on `stream` (a raw hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) from
        now on runs after every conversion queued on this context so far - on the device, no host wait"""
        if not self._lib.horizonator_amd_stream_waits_for_outputs(C.byref(self._ctx), C.c_void_p(int(stream))):
            raise RuntimeError("horizonator_amd_stream_waits_for_outputs() failed")

    def waits_for_stream(self, stream):
        """conversions queued on this context from now on run after everything queued on `stream` so far"""
        if not self._lib.horizonator_amd_waits_for_stream(C.byref(self._ctx), C.c_void_p(int(stream))):
            raise RuntimeError("horizonator_amd_waits_for_stream() failed")

    def set_sector(self, col0, col1):
        if not self._lib.horizonator_amd_set_sector(C.byref(self._ctx), int(col0), int(col1)):
            raise RuntimeError("horizonator_amd_set_sector() failed")
        self._sector = (int(col0), int(col1))

    def set_raster(self, which):
        if not self._lib.horizonator_amd_set_raster(C.byref(self._ctx), int(which)):
            raise RuntimeError("horizonator_amd_set_raster() failed")

    def options(self):
        """the context's tunables (include/hz_hip.h: hz_options_t) as a dict"""
        o = Options()
        if not self._lib.horizonator_amd_get_options(C.byref(self._ctx), C.byref(o)):
            raise RuntimeError("horizonator_amd_get_options() failed")
        return {n: getattr(o, n) for n, _ in Options._fields_}

    def set_options(self, **kw):
        """change some of the tunables, e.g. set_options(rounds=1, host_sectors=4): none changes a byte of a result"""
        o = Options()
        if not self._lib.horizonator_amd_get_options(C.byref(self._ctx), C.byref(o)):
            raise RuntimeError("horizonator_amd_get_options() failed")
        for k, v in kw.items():
            if k not in dict(Options._fields_):
                raise TypeError("no such option: %s" % k)
            setattr(o, k, int(v))
        if not self._lib.horizonator_amd_set_options(C.byref(self._ctx), C.byref(o)):
            raise RuntimeError("horizonator_amd_set_options() failed")

    def set_profiling(self, on=True):
        self._lib.horizonator_amd_set_profiling(C.byref(self._ctx), bool(on))

    def last_times(self):
        t = Times()
        if not self._lib.horizonator_amd_last_times(C.byref(self._ctx), C.byref(t)):
            return None
        return {n: getattr(t, n) for n, _ in Times._fields_}

    def last_plan(self):
        """what the last draw was: {"rounds": 1 | 2, "coarse_depth": its second round kept coarse depth (zoomed views, the
        draws of a series), "reach_cells": the first round's reach, "work_list": only the strips behind the drawn columns
        were launched, "vertex_cache": its vertices came from the vertex cache} (include/hz_hip.h: hz_hip_last_plan)"""
        out = (C.c_int * 5)()
        if self._lib.hz_hip_last_plan(self._lib.horizonator_amd_device(C.byref(self._ctx)), out) != 0:
            raise RuntimeError("hz_hip_last_plan() failed")
        return {"rounds": int(out[0]), "coarse_depth": bool(out[1]), "reach_cells": int(out[2]), "work_list": bool(out[3]),
                "vertex_cache": bool(out[4])}

    def view(self):
        v = View()
        if not self._lib.horizonator_amd_get_view(C.byref(self._ctx), C.byref(v)):
            raise RuntimeError("horizonator_amd_get_view() failed")
        return {n: getattr(v, n) for n, _ in View._fields_}

    def mosaic(self):
        N = 2 * self.radius_cells
        m = np.empty((N, N), np.int16)
        if not self._lib.horizonator_amd_get_mosaic(C.byref(self._ctx), m.ctypes.data):
            raise RuntimeError("horizonator_amd_get_mosaic() failed")
        return m

    def link_cells(self, cell_width=14, cell_height=14, cut_off_bottom_px=0):
        """lat/lon under the centre of every cell of the last render that shows
        terrain (what the reference's annotator turns into map links, reference
        annotator.c:228-264), computed on the device: (lat, lon) float32[ny,nx],
        NaN where the cell shows sky."""
        nx, ny = C.c_int(), C.c_int()
        ctx = C.byref(self._ctx)
        if not self._lib.horizonator_amd_link_cells_size(ctx, cell_width, cell_height, cut_off_bottom_px,
                                                         C.byref(nx), C.byref(ny)):
            raise RuntimeError("horizonator_amd_link_cells_size() failed")
        lat = np.full((ny.value, nx.value), np.nan, np.float32)
        lon = np.full((ny.value, nx.value), np.nan, np.float32)
        if nx.value and ny.value:
            if not self._lib.horizonator_amd_link_cells(ctx, cell_width, cell_height, cut_off_bottom_px,
                                                        lat.ctypes.data, lon.ctypes.data):
                raise RuntimeError("horizonator_amd_link_cells() failed")
        return lat, lon

    def poi_visibility(self, pois, cut_off_bottom_px=0):
        """pois: float32[n,3] = lat, lon, elevation (m).  Which of them show in
        the last render and where their label crosshair goes (reference
        annotator.c:280-348), computed on the device: (visible uint8[n],
        x float32[n], y float32[n])."""
        pois = np.ascontiguousarray(pois, np.float32).reshape(-1, 3)
        n = pois.shape[0]
        vis = np.zeros(n, np.uint8)
        x = np.zeros(n, np.float32)
        y = np.zeros(n, np.float32)
        if not self._lib.horizonator_amd_poi_visibility(C.byref(self._ctx), cut_off_bottom_px, pois.ctypes.data, n,
                                                        vis.ctypes.data, x.ctypes.data, y.ctypes.data):
            raise RuntimeError("horizonator_amd_poi_visibility() failed")
        return vis, x, y

    def pick(self, x, y):
        """(lat, lon) of the terrain under image pixel (x, y), or None for sky
        (reference horizonator.h:145-152)."""
        lat, lon = C.c_float(), C.c_float()
        if not self._lib.horizonator_pick(C.byref(self._ctx), C.byref(lat), C.byref(lon), int(x), int(y)):
            return None
        return lat.value, lon.value
