"""ctypes binding of libhorizonator.so (include/horizonator.h, horizonator_amd.h, hz_hip.h).

The shared library is the product; this module only declares its C-ABI.  There
is no Python or CPU fallback: if the library is missing, importing fails and
says how to build it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HORIZONATOR_AMD_LIB: another build of the same library (tests/test_sanitizers.py loads the
# address/undefined-behaviour-sanitized host code through it)
LIB_PATH = os.environ.get("HORIZONATOR_AMD_LIB") or os.path.join(_HERE, "libhorizonator.so")
DEMGEN_PATH = os.path.join(_HERE, "libhzdemgen.so")
# the library's own sources built with -DHZ_SELFTEST (include/hz_selftest.h): device-side self-checks and
# diagnostics entry points on top of everything libhorizonator.so exports; tests and tools only
SELFTEST_PATH = os.path.join(_HERE, "libhorizonator_selftest.so")

MAX_NDEMS_IJ = 4


class DemContext(C.Structure):
    """horizonator_dem_context_t (include/dem.h; reference dem.h:10-29)"""
    _fields_ = [
        ("dems", (C.c_void_p * MAX_NDEMS_IJ) * MAX_NDEMS_IJ),
        ("mmap_sizes", (C.c_size_t * MAX_NDEMS_IJ) * MAX_NDEMS_IJ),
        ("mmap_fd", (C.c_int * MAX_NDEMS_IJ) * MAX_NDEMS_IJ),
        ("origin_dem_lon_lat", C.c_int * 2),
        ("origin_dem_cellij", C.c_int * 2),
        ("Ndems_ij", C.c_int * 2),
        ("radius_cells", C.c_int),
        ("cells_per_deg", C.c_int),
    ]


class _Offscreen(C.Structure):
    _fields_ = [
        ("inited", C.c_bool),
        ("frameBufID", C.c_uint32),
        ("renderBufID", C.c_uint32),
        ("depthBufID", C.c_uint32),
        ("width", C.c_int),
        ("height", C.c_int),
    ]


class Context(C.Structure):
    """horizonator_context_t (include/horizonator.h; reference horizonator.h:13-53)"""
    _fields_ = (
        [("Ntriangles", C.c_int), ("render_texture", C.c_bool), ("use_glut", C.c_bool),
         ("glut_window", C.c_int)]
        + [(name, C.c_int32) for name in (
            "uniform_aspect", "uniform_az_deg0", "uniform_az_deg1",
            "uniform_viewer_cell_i", "uniform_viewer_cell_j", "uniform_viewer_z",
            "uniform_viewer_lat", "uniform_cos_viewer_lat",
            "uniform_texturemap_lon0", "uniform_texturemap_lon1",
            "uniform_texturemap_dlat0", "uniform_texturemap_dlat1", "uniform_texturemap_dlat2",
            "uniform_znear", "uniform_zfar", "uniform_znear_color", "uniform_zfar_color")]
        + [("program", C.c_uint32), ("viewer_lat", C.c_float), ("viewer_lon", C.c_float),
           ("dems", DemContext), ("offscreen", _Offscreen)]
    )


class View(C.Structure):
    """hz_view_t (include/hz_hip.h)"""
    _fields_ = [(n, C.c_float) for n in (
        "viewer_cell_i", "viewer_cell_j", "viewer_z", "cos_viewer_lat", "deg_per_cell",
        "az_deg0", "az_deg1", "aspect", "znear", "zfar", "znear_color", "zfar_color")]


class Options(C.Structure):
    """hz_options_t (include/hz_hip.h): the tunables of a context"""
    _fields_ = [(n, C.c_int) for n in (
        "serial", "rounds", "near_cells", "coarse_depth", "tiles", "tile_list", "adapt", "adapt_hi", "pretest_march",
        "worklists", "fast_math", "resolve_clears", "queue_capacity", "host_dense", "host_sectors", "host_times", "vertex_cache")]


class Times(C.Structure):
    """hz_times_t (include/hz_hip.h)"""
    _fields_ = [(n, C.c_float) for n in ("clear_ms", "raster_ms", "big_ms", "resolve_ms", "total_ms", "near_ms")]


class Window(C.Structure):
    """horizonator_amd_window_t (include/horizonator_amd.h)"""
    _fields_ = [("cells_per_deg", C.c_int), ("radius_cells", C.c_int),
                ("origin_tile", C.c_int * 2), ("origin_cell", C.c_int * 2)]


class TexParams(C.Structure):
    """hz_texparams_t (include/hz_hip.h)"""
    _fields_ = [(n, C.c_float) for n in ("viewer_lat_rad", "origin_cell_lon_deg", "origin_cell_lat_deg",
                                         "lon0", "lon1", "dlat0", "dlat1", "dlat2")] + \
               [(n, C.c_int32) for n in ("ntiles_x", "ntiles_y", "lowest_x", "lowest_y", "tex_w", "tex_h")]


RASTER_AUTO, RASTER_SCATTER, RASTER_MARCH = 0, 1, 2

_lib = None
_selftest = None


def load():
    """dlopen libhorizonator.so and declare every prototype.  Raises if absent."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH, selftest=os.path.basename(LIB_PATH) == os.path.basename(SELFTEST_PATH))
    return _lib


def load_selftest():
    """dlopen libhorizonator_selftest.so: the same library with the self-checks and diagnostics of
    include/hz_selftest.h.  A context made through it is its own (the two libraries share no state)."""
    global _selftest
    if _selftest is None:
        _selftest = _open(SELFTEST_PATH, selftest=True)
    return _selftest


def _open(path, selftest):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make -C horizonator_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "horizonator_amd has no fallback implementation.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL if not selftest else C.RTLD_LOCAL)
    P = C.POINTER
    ctxp = P(Context)
    f, i, b, d = C.c_float, C.c_int, C.c_bool, C.c_double
    vp = C.c_void_p

    def sig(name, res, *args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("horizonator_init", b, ctxp, f, f, P(f), i, i, i, f, b, b, b,
        C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, b)
    sig("horizonator_deinit", None, ctxp)
    sig("horizonator_resized", b, ctxp, i, i)
    sig("horizonator_pan_zoom", b, ctxp, f, f)
    sig("horizonator_move", b, ctxp, P(f), f, f)
    sig("horizonator_set_zextents", b, ctxp, f, f, f, f)
    sig("horizonator_redraw", b, ctxp)
    sig("horizonator_pick", b, ctxp, P(f), P(f), i, i)
    sig("horizonator_render_offscreen", b, ctxp, vp, vp)
    sig("horizonator_x_from_az", b, P(d), P(d), d, d, d, i)
    sig("horizonator_project", b, P(d), P(d), P(d), d, d, d, d, d, d, d, d, d, i, i)
    sig("horizonator_unproject", b, P(f), P(f), i, i, d, d, d, d, d, d, d, i, i)

    sig("horizonator_dem_init", b, P(DemContext), f, f, i, f, C.c_char_p, b)
    sig("horizonator_dem_deinit", None, P(DemContext))
    sig("horizonator_dem_sample", C.c_int16, P(DemContext), i, i)
    sig("horizonator_dem_bounds_latlon_deg", None, P(DemContext), P(f), P(f), P(f), P(f))

    sig("horizonator_amd_get_window", b, ctxp, P(Window))
    sig("horizonator_amd_init_from_mosaic", b, ctxp, f, f, P(f), i, i, P(Window), vp)
    sig("horizonator_amd_render", b, ctxp, vp, vp, vp, vp)
    sig("horizonator_amd_render_device", b, ctxp, vp, vp, vp, vp)
    sig("horizonator_amd_render_batch", b, ctxp, i, vp, vp, vp, vp, vp)
    sig("horizonator_amd_render_packed", b, ctxp, vp)
    sig("horizonator_amd_resolve_packed", b, ctxp, vp, i, i, i, vp, vp)
    sig("horizonator_amd_resolve_packed_strips", b, ctxp, i, vp, i, vp, vp, vp, vp)
    sig("horizonator_amd_render_sparse", b, ctxp, vp, i)
    sig("horizonator_amd_resolve_sparse_strips", b, ctxp, i, vp, i, vp, vp, vp, vp)
    sig("horizonator_amd_sync", b, ctxp)
    sig("horizonator_amd_texture_layout", b, ctxp, P(i), P(i), P(i), P(i))
    sig("horizonator_amd_set_texture", b, ctxp, vp)
    sig("horizonator_amd_set_sector", b, ctxp, i, i)
    sig("horizonator_amd_set_raster", b, ctxp, i)
    sig("horizonator_amd_set_profiling", b, ctxp, b)
    sig("horizonator_amd_render_begin", b, ctxp, vp, vp)
    sig("horizonator_amd_render_end", b, ctxp)
    sig("horizonator_amd_get_options", b, ctxp, P(Options))
    sig("horizonator_amd_set_options", b, ctxp, P(Options))
    sig("horizonator_amd_last_times", b, ctxp, P(Times))
    sig("horizonator_amd_get_view", b, ctxp, P(View))
    sig("horizonator_amd_device", vp, ctxp)
    sig("horizonator_amd_get_mosaic", b, ctxp, vp)
    sig("horizonator_amd_build_id", C.c_char_p)
    sig("horizonator_amd_link_cells_size", b, ctxp, i, i, i, P(i), P(i))
    sig("horizonator_amd_link_cells", b, ctxp, i, i, i, vp, vp)
    sig("horizonator_amd_poi_visibility", b, ctxp, i, vp, i, vp, vp, vp)

    sig("hz_hip_device_count", i)
    sig("hz_hip_create", vp, i, i, i, i)
    sig("hz_hip_destroy", None, vp)
    sig("hz_hip_upload_mosaic", i, vp, vp)
    sig("hz_hip_download_mosaic", i, vp, vp)
    sig("hz_hip_ingest_tiles", i, vp, vp, i, i, i, i, i)
    sig("hz_hip_set_sector", i, vp, i, i)
    sig("hz_hip_set_raster", i, vp, i)
    sig("hz_hip_set_profiling", i, vp, i)
    sig("hz_hip_get_options", i, vp, P(Options))
    sig("hz_hip_set_options", i, vp, P(Options))
    sig("hz_hip_set_texture", i, vp, P(TexParams), vp)
    sig("hz_hip_pack", i, vp, vp)
    sig("hz_hip_resolve_packed", i, vp, P(View), vp, vp, i, i, i, vp, vp)
    sig("hz_hip_pack_sparse", i, vp, vp, i)
    sig("hz_hip_resolve_sparse", i, vp, P(View), vp, vp, i, i, i, vp, vp)
    sig("hz_hip_resolve_sparse_strips", i, vp, P(View), vp, i, vp, i, vp, vp, vp, vp)
    sig("hz_hip_draw", i, vp, P(View))
    sig("hz_hip_resolve", i, vp, P(View), vp, vp, vp, vp, vp)
    sig("hz_hip_resolve_to_host", i, vp, P(View), vp, vp, vp, vp, vp)
    sig("hz_hip_render_to_host", i, vp, P(View), vp, vp, vp, vp, vp)
    sig("hz_hip_host_begin", i, vp, P(View), vp, vp, vp, vp, vp)
    sig("hz_hip_host_end", i, vp)
    sig("hz_hip_host_prepare", i, vp, i, i, i, i, vp)
    sig("hz_hip_read_depth", i, vp, i, i, P(C.c_uint32))
    sig("hz_hip_link_cells", i, vp, P(View), vp, vp, vp, vp, d, d, d, i, i, i, i, vp, vp)
    sig("hz_hip_poi_visibility", i, vp, P(View), vp, i, vp, i, vp, vp, vp)
    sig("hz_hip_sync", i, vp)
    sig("hz_hip_last_times", i, vp, P(Times))
    sig("hz_hip_stream", vp, vp)
    sig("hz_hip_wait_outputs", i, vp, vp)
    sig("hz_hip_wait_for", i, vp, vp)
    sig("horizonator_amd_stream_waits_for_outputs", b, ctxp, vp)
    sig("horizonator_amd_waits_for_stream", b, ctxp, vp)
    sig("hz_hip_last_plan", i, vp, vp)
    sig("hz_hip_last_queue_counts", i, vp, vp)
    sig("hz_hip_last_error", C.c_char_p)
    if selftest:
        sig("hz_hip_check_fastmath", i, i, i, C.c_uint64, C.c_uint64, P(C.c_uint64), vp)
        sig("hz_hip_check_exactness", i, i, i, C.c_uint64, C.c_uint64, i, i, i, i, P(C.c_uint64))
        sig("hz_hip_debug_bigqueue", i, vp, i, vp, i, vp)
        sig("hz_hip_debug_wave_timing", i, vp, P(View), vp, C.c_size_t, vp)
        sig("hz_hip_debug_worklist", C.c_long, i, i, i, P(View), i, i, i, vp, C.c_size_t)
    return lib


# every symbol include/*.h declares; tests check the library exports all of them
DECLARED_SYMBOLS = [
    # include/horizonator.h
    "horizonator_init", "horizonator_deinit", "horizonator_resized", "horizonator_pan_zoom",
    "horizonator_move", "horizonator_set_zextents", "horizonator_redraw", "horizonator_pick",
    "horizonator_render_offscreen", "horizonator_x_from_az", "horizonator_project",
    "horizonator_unproject",
    # include/dem.h
    "horizonator_dem_init", "horizonator_dem_deinit", "horizonator_dem_sample",
    "horizonator_dem_bounds_latlon_deg",
    # include/horizonator_amd.h
    "horizonator_amd_get_window", "horizonator_amd_init_from_mosaic",
    "horizonator_amd_render", "horizonator_amd_render_device", "horizonator_amd_render_batch",
    "horizonator_amd_render_packed", "horizonator_amd_resolve_packed",
    "horizonator_amd_resolve_packed_strips", "horizonator_amd_render_sparse",
    "horizonator_amd_resolve_sparse_strips",
    "horizonator_amd_sync", "horizonator_amd_stream_waits_for_outputs", "horizonator_amd_waits_for_stream", "horizonator_amd_texture_layout", "horizonator_amd_set_texture",
    "horizonator_amd_set_sector", "horizonator_amd_set_raster", "horizonator_amd_set_profiling",
    "horizonator_amd_get_options", "horizonator_amd_set_options", "horizonator_amd_render_begin", "horizonator_amd_render_end",
    "horizonator_amd_last_times", "horizonator_amd_get_view", "horizonator_amd_device",
    "horizonator_amd_get_mosaic", "horizonator_amd_link_cells_size", "horizonator_amd_link_cells",
    "horizonator_amd_poi_visibility", "horizonator_amd_build_id",
    # include/hz_hip.h
    "hz_hip_device_count", "hz_hip_create", "hz_hip_destroy", "hz_hip_upload_mosaic",
    "hz_hip_download_mosaic", "hz_hip_ingest_tiles", "hz_hip_set_sector", "hz_hip_set_raster",
    "hz_hip_set_profiling", "hz_hip_get_options", "hz_hip_set_options", "hz_hip_set_texture", "hz_hip_pack", "hz_hip_resolve_packed", "hz_hip_pack_sparse", "hz_hip_resolve_sparse", "hz_hip_resolve_sparse_strips", "hz_hip_draw", "hz_hip_resolve", "hz_hip_resolve_to_host", "hz_hip_render_to_host", "hz_hip_host_begin", "hz_hip_host_end", "hz_hip_host_prepare",
    "hz_hip_read_depth", "hz_hip_link_cells", "hz_hip_poi_visibility", "hz_hip_sync", "hz_hip_last_times", "hz_hip_stream", "hz_hip_wait_outputs", "hz_hip_wait_for", "hz_hip_last_plan", "hz_hip_last_queue_counts", "hz_hip_last_error",
]
# include/hz_selftest.h: what libhorizonator_selftest.so exports on top of those (and libhorizonator.so must not)
SELFTEST_SYMBOLS = ["hz_hip_check_fastmath", "hz_hip_check_exactness", "hz_hip_debug_bigqueue", "hz_hip_debug_wave_timing",
                    "hz_hip_debug_worklist"]


def load_demgen():
    """libhzdemgen.so: deterministic synthetic SRTM tiles (tools/demgen.c)."""
    if not os.path.exists(DEMGEN_PATH):
        raise ImportError(f"{DEMGEN_PATH} is missing: build it with `make -C horizonator_amd/csrc`")
    lib = C.CDLL(DEMGEN_PATH)
    lib.hz_demgen_write_region.restype = C.c_int
    lib.hz_demgen_write_region.argtypes = [C.c_char_p] + [C.c_int] * 6
    lib.hz_demgen_write_tile.restype = C.c_int
    lib.hz_demgen_write_tile.argtypes = [C.c_char_p] + [C.c_int] * 4
    lib.hz_demgen_tile_values.restype = None
    lib.hz_demgen_tile_values.argtypes = [C.c_void_p] + [C.c_int] * 4
    return lib
