"""the C-ABI library loads and exports every symbol include/*.h declares (no GPU needed)"""
import ctypes as C
import os
import re

from horizonator_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_in_headers():
    names = set()
    for hdr in ("horizonator.h", "dem.h", "horizonator_amd.h", "hz_hip.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b((?:horizonator|hz_hip)_[a-z0-9_]+)\s*\(", text))
    names.discard("horizonator_context_isvalid")      # static inline in the header
    return names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_in_headers()
    assert declared == set(_lib.DECLARED_SYMBOLS), declared ^ set(_lib.DECLARED_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_self_checks_and_diagnostics_live_in_a_library_of_their_own():
    """include/hz_selftest.h -> libhorizonator_selftest.so (the library's sources + the device-side self-checks and
    the diagnostics of tools/): it exports everything the product does plus those; the product exports none of them"""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "hz_selftest.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(hz_hip_[a-z0-9_]+)\s*\(", text))
    assert declared == set(_lib.SELFTEST_SYMBOLS), declared ^ set(_lib.SELFTEST_SYMBOLS)
    product, selftest = _lib.load(), _lib.load_selftest()
    for name in sorted(declared):
        assert hasattr(selftest, name), name
        assert not hasattr(product, name), f"{name}: a self-check / diagnostics entry point in the library that ships"
    for name in _lib.DECLARED_SYMBOLS:
        assert hasattr(selftest, name), name


def test_rccl_library_exports_what_its_header_declares():
    """include/horizonator_rccl.h -> libhorizonator_rccl.so (the exchange steps for a C caller; a
    library of its own so that libhorizonator.so does not depend on RCCL)"""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "horizonator_rccl.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(horizonator_rccl_[a-z0-9_]+)\s*\(", text))
    assert declared == {"horizonator_rccl_broadcast_mosaic", "horizonator_rccl_gather_strips", "horizonator_rccl_render_series"}
    _lib.load()                                     # libhorizonator.so first: the rccl library links against it
    lib = C.CDLL(os.path.join(ROOT, "horizonator_amd", "libhorizonator_rccl.so"))
    for name in declared:
        assert hasattr(lib, name), name


def test_context_layout_matches_reference_abi():
    # numbers printed by a program compiled against the REFERENCE's own
    # horizonator.h / dem.h (gcc 11, x86-64): sizeof(ctx), offsetof program,
    # viewer_lat, dems, offscreen, sizeof(dem ctx) = 472 80 84 96 448 352
    assert C.sizeof(_lib.Context) == 472
    assert _lib.Context.program.offset == 80
    assert _lib.Context.viewer_lat.offset == 84
    assert _lib.Context.dems.offset == 96
    assert _lib.Context.offscreen.offset == 448
    assert C.sizeof(_lib.DemContext) == 352


def test_context_layout_matches_c_compiler(tmp_path):
    """the ctypes mirror agrees with what gcc makes of include/horizonator.h"""
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "horizonator.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(horizonator_context_t),'
        'offsetof(horizonator_context_t,program), offsetof(horizonator_context_t,viewer_lat),'
        'offsetof(horizonator_context_t,dems), offsetof(horizonator_context_t,offscreen),'
        'sizeof(horizonator_dem_context_t)); return 0;}\n')
    exe = tmp_path / "layout"
    import subprocess
    subprocess.check_call(["gcc", "-std=gnu99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(_lib.Context), _lib.Context.program.offset, _lib.Context.viewer_lat.offset,
                   _lib.Context.dems.offset, _lib.Context.offscreen.offset, C.sizeof(_lib.DemContext)]


def test_additional_structs_match_c_compiler(tmp_path):
    """the ctypes mirrors of the build-side structs (hz_hip.h, horizonator_amd.h) agree with gcc"""
    import subprocess
    names = [("hz_view_t", _lib.View), ("hz_times_t", _lib.Times), ("hz_texparams_t", _lib.TexParams),
             ("horizonator_amd_window_t", _lib.Window)]
    body = "".join(f'printf("%zu ", sizeof({n}));' for n, _ in names)
    last = [(n, t._fields_[-1][0]) for n, t in names]
    body += "".join(f'printf("%zu ", offsetof({n}, {f}));' for n, f in last)
    src = tmp_path / "structs.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "horizonator_amd.h"\n'
                   'int main(void){' + body + 'return 0;}\n')
    exe = tmp_path / "structs"
    subprocess.check_call(["gcc", "-std=gnu99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(t) for _, t in names] + [getattr(t, t._fields_[-1][0]).offset for _, t in names]
    assert got == want


def test_no_gpu_means_loud_failure():
    """without a HIP device the product refuses to run; it never falls back to a CPU path"""
    lib = _lib.load()
    if lib.hz_hip_device_count() > 0:
        return
    import pytest
    import horizonator_amd
    import hzutil
    d = hzutil.dem_dir_for(hzutil.VIEW_LAT, hzutil.VIEW_LON, 32)
    with pytest.raises(RuntimeError):
        horizonator_amd.horizonator(hzutil.VIEW_LAT, hzutil.VIEW_LON, 64, 16, dir_dems=d, render_radius_cells=32)


def test_product_does_not_touch_the_oracle():
    """nothing under horizonator_amd/ or include/ may import, include or link oracle/"""
    bad = []
    for base in ("horizonator_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dirpath or "__pycache__" in dirpath:
                continue
            for fn in files:
                if not fn.endswith((".c", ".h", ".hip", ".py", "Makefile")):
                    continue
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                if re.search(r"(#include\s*[\"<].*oracle|import\s+oracle|from\s+oracle|liboracle|orc_)", text):
                    bad.append(os.path.join(dirpath, fn))
    assert not bad, bad


def test_series_struct_of_the_rccl_header_matches_c_compiler(tmp_path):
    """sharding._Series (what RcclSeries hands to horizonator_rccl_render_series) agrees with what gcc makes of
    horizonator_rccl_series_t: size and the offset of every field"""
    import subprocess
    from horizonator_amd.sharding import _Series
    fields = [f for f, _ in _Series._fields_]
    body = 'printf("%zu ", sizeof(horizonator_rccl_series_t));' + "".join(f'printf("%zu ", offsetof(horizonator_rccl_series_t, {f}));' for f in fields)
    src = tmp_path / "series.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "horizonator_rccl.h"\nint main(void){' + body + 'return 0;}\n')
    exe = tmp_path / "series"
    subprocess.check_call(["gcc", "-std=gnu99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(_Series)] + [getattr(_Series, f).offset for f in fields]


def test_series_rejects_bad_arguments_without_a_gpu():
    """horizonator_rccl_render_series checks its arguments before it touches the device: no context, no communicator,
    a rank outside the communicator, too many slots, fewer words than a header - -1 each, nothing queued"""
    from horizonator_amd.sharding import _Series
    _lib.load()
    lib = C.CDLL(os.path.join(ROOT, "horizonator_amd", "libhorizonator_rccl.so"))
    lib.horizonator_rccl_render_series.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Series), C.c_long, C.c_int, C.c_int]
    ctx = _lib.Context()
    fake_comm = C.c_void_p(1)
    strips = (C.c_void_p * 2)(8, 16)
    col0 = (C.c_int * 1)(0)
    ncols = (C.c_int * 1)(100)

    def series(**kw):
        v = dict(rank=0, world=1, rotate=0, nslots=2, d_strips=C.cast(strips, C.POINTER(C.c_void_p)), d_bins=C.cast(strips, C.POINTER(C.c_void_p)),
                 words=1000, header_words=10, mask_stride=4, col0=C.cast(col0, C.POINTER(C.c_int)), ncols=C.cast(ncols, C.POINTER(C.c_int)),
                 d_image=None, d_ranges=None, stream=None)
        v.update(kw)
        return _Series(**v)

    call = lib.horizonator_rccl_render_series
    assert call(None, fake_comm, C.byref(series()), 0, 1, 0) == -1
    assert call(C.byref(ctx), None, C.byref(series()), 0, 1, 0) == -1
    assert call(C.byref(ctx), fake_comm, None, 0, 1, 0) == -1
    assert call(C.byref(ctx), fake_comm, C.byref(series(rank=1)), 0, 1, 0) == -1
    assert call(C.byref(ctx), fake_comm, C.byref(series(nslots=5)), 0, 1, 0) == -1
    assert call(C.byref(ctx), fake_comm, C.byref(series(words=10)), 0, 1, 0) == -1
    assert call(C.byref(ctx), fake_comm, C.byref(series(d_bins=None)), 0, 1, 0) == -1       # rank 0 gathers: it needs bins
    assert call(C.byref(ctx), fake_comm, C.byref(series()), -1, 1, 0) == -1


def test_result_memory_of_the_python_mirror_is_recycled_only_when_dropped(monkeypatch):
    """render() makes its results of memory the caller has dropped (horizonator_amd._ResultMemory): never of memory an
    array or a view of one still names, two calls' worth kept at most, HZ_PY_RECYCLE=0 = new arrays every time"""
    import numpy as np
    import horizonator_amd
    specs = [((6, 10, 3), np.uint8), ((6, 10), np.float32)]
    m = horizonator_amd._ResultMemory()
    image, ranges = m.take(specs)
    assert image.shape == (6, 10, 3) and image.dtype == np.uint8 and image.flags.c_contiguous and image.flags.writeable
    assert ranges.shape == (6, 10) and ranges.dtype == np.float32 and ranges.flags.c_contiguous
    first = {image.ctypes.data, ranges.ctypes.data}
    image[:] = 7
    image2, ranges2 = m.take(specs)                     # the first results are still held: other memory
    assert not {image2.ctypes.data, ranges2.ctypes.data} & first and (image == 7).all()
    row = image[2]                                      # a view keeps the memory as well
    del image, ranges
    image3, ranges3 = m.take(specs)
    assert image3.ctypes.data not in first and ranges3.ctypes.data in first
    assert (row == 7).all()
    del row, image2, ranges2, image3, ranges3
    seen = set()
    for _ in range(8):                                  # `image, ranges = h.render(...)` in a loop: the names hold one set during the next call
        image, ranges = m.take(specs)
        seen |= {image.ctypes.data, ranges.ctypes.data}
    assert len(seen) == 4 and len(m._bufs) <= 4
    # another size (a sector was set): nothing of the old size is handed out, and it does not pile up
    small = m.take([((6, 4, 3), np.uint8)])[0]
    assert small.shape == (6, 4, 3) and len(m._bufs) <= 2
    monkeypatch.setenv("HZ_PY_RECYCLE", "0")
    off = horizonator_amd._ResultMemory()
    a = off.take(specs)[0]
    p = a.ctypes.data
    assert a.flags.owndata
    del a
    assert off._bufs == []
