"""The reference's own command-line program on the MI355X: standalone.c + annotator.c, compiled in the
build container where they lie in /root/reference and linked UNCHANGED against include/ + libhorizonator.so
(tests/caller_stubs/Makefile; stand-ins only for the CLI's own dependencies), run here as a child process.
Its --image out.png path is horizonator_init / set_zextents / pan_zoom / render_offscreen (reference
standalone.c:433-460) and then FreeImage, whose stand-in writes the bytes it is handed to the file: they
must be the bytes this library's Python mirror renders for the same view.  The .pdf path additionally
runs the reference's annotate() (annotator.c: horizonator_unproject / _project over the range image)."""
import os
import subprocess

import numpy as np
import pytest

import hzutil

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "caller_stubs", "_built", "standalone_ref")


def _read_raw(path):
    with open(path, "rb") as f:
        tag, w, h, bpp = f.readline().split()
        assert tag == b"HZRAW"
        w, h, bpp = int(w), int(h), int(bpp)
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, bpp)


def test_the_reference_cli_renders_through_this_library(tmp_path):
    import horizonator_amd
    if not os.path.exists(EXE):
        # the binary is built where /root/reference is (the build container) and travels with the tree, untracked: on a
        # box that got neither, this test cannot run - and says so where a -q run shows it (pytest.ini: -rs)
        why = (f"NOT RUN: {os.path.relpath(EXE, ROOT)} is missing - the reference's standalone.c is linked against this library only "
               "where /root/reference exists (python -c 'import __graft_entry__ as g; g.build()' in the build container)")
        print("\n" + why, flush=True)
        pytest.skip(why)
    # the binary is the one linked against THIS build of the library (it is untracked and travels with the tree)
    from horizonator_amd import _lib
    import ctypes
    lib = _lib.load()
    lib.horizonator_amd_build_id.restype = ctypes.c_char_p
    build_id = lib.horizonator_amd_build_id().decode()
    r = subprocess.run([EXE, "--help"], capture_output=True, text=True, timeout=60, env=dict(os.environ, HZ_SHOW_BUILD_ID="1"))
    assert f"linked against libhorizonator build {build_id}" in r.stderr, (build_id, r.stderr[-300:])
    lat, lon, W, H, zfar, azc, azr = hzutil.VIEW_LAT, hzutil.VIEW_LON, 1200, 300, 20000.0, 35.0, 70.0
    # radius given in metres (reference standalone.c:438: radius_cells = -1, radius_m = zfar)
    R = int(round(zfar / (6371000.0 * np.pi / 180.0 * np.cos(np.radians(np.float32(lat))) / 1200)))
    dems = hzutil.dem_dir_for(lat, lon, R + 8)
    out = tmp_path / "out.png"
    r = subprocess.run([EXE, "--width", str(W), "--height", str(H), "--image", str(out), "--dirdems", dems, "--zfar", str(zfar),
                        str(lat), str(lon), str(azc), str(azr)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and out.exists(), r.stderr[-2000:]
    got = _read_raw(str(out))
    assert got.shape == (H, W, 3)
    # the same view through the Python mirror: the CLI's azimuths are those of the first and last pixel CENTRES,
    # widened by half a pixel in float32 (reference standalone.c:400-404)
    rad = np.float32(azr)
    rad = rad + np.float32(2.0 * float(rad) / np.float32(W - 1)) / np.float32(2.0)
    h = horizonator_amd.horizonator(lat, lon, W, H, dir_dems=dems, render_radius_m=zfar)
    try:
        image, ranges = h.render(float(np.float32(azc) - rad), float(np.float32(azc) + rad), zfar=zfar)
    finally:
        h.close()
    assert (image[..., 2] > 0).mean() > 0.05                     # terrain is in view
    assert np.array_equal(got, image), f"{(got != image).any(axis=2).sum()} pixels differ"
    # ... and the annotated output: annotate() runs over the range image (cairo is a stand-in: nothing is drawn)
    pdf = tmp_path / "out.pdf"
    r = subprocess.run([EXE, "--width", str(W), "--height", str(H), "--image", str(pdf), "--dirdems", dems, "--zfar", str(zfar),
                        str(lat), str(lon), str(azc), str(azr)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
