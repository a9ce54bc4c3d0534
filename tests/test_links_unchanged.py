"""The drop-in boundary, kept: the reference's own Python wrapper, horizonator-pywrap.c, is
compiled WHERE IT LIES in /root/reference (nothing is copied) against this repo's include/ and
linked against libhorizonator.so - unchanged, as INTEGRATION.md tells a maintainer to.  Every
horizonator_* symbol it leaves undefined must be one this library exports, the built module
must import, and its type must carry the reference's method and docstrings.  Skipped where the
reference is absent (the GPU box)."""
import os
import subprocess
import sys
import sysconfig

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "horizonator_amd")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "horizonator-pywrap.c")),
                                reason="the reference's sources are not on this machine")


def _docstring_header(src, dst):
    # the reference generates these with its build system's string-literal rule (Makefile:22-24
    # is the same rule for the shaders): every line becomes a C string literal ending in \n
    with open(src) as f, open(dst, "w") as g:
        for line in f.read().splitlines():
            g.write('"' + line.replace("\\", "\\\\").replace('"', '\\"') + '\\n"\n')


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    out = tmp_path_factory.mktemp("pywrap")
    for d in ("horizonator", "render"):
        _docstring_header(os.path.join(REF, d + ".docstring"), str(out / (d + ".docstring.h")))
    so = str(out / ("horizonator" + sysconfig.get_config_var("EXT_SUFFIX")))
    cmd = ["gcc", "-std=gnu99", "-shared", "-fPIC", "-O1",
           "-I" + os.path.join(ROOT, "include"), "-I" + str(out),
           "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(),
           os.path.join(REF, "horizonator-pywrap.c"), "-o", so,
           "-L" + LIBDIR, "-lhorizonator", "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "the reference's horizonator-pywrap.c no longer builds against include/:\n" + r.stderr
    return so, str(out)


def test_every_horizonator_symbol_the_wrapper_needs_is_exported(built):
    so, _ = built
    undefined = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--undefined-only", so], text=True).splitlines()
                 if "horizonator" in l}
    exported = {l.split()[-1] for l in subprocess.check_output(
        ["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libhorizonator.so")], text=True).splitlines()}
    # reference horizonator-pywrap.c:107-117 (init), :214-232 (pan_zoom, move, set_zextents), :252-260 (render_offscreen), dealloc
    assert undefined == {"horizonator_init", "horizonator_deinit", "horizonator_pan_zoom", "horizonator_move",
                         "horizonator_set_zextents", "horizonator_render_offscreen"}
    assert undefined <= exported


def test_the_built_module_imports_and_is_the_reference_type(built):
    so, outdir = built
    code = ("import sys; sys.path.insert(0, %r); import horizonator as m; t = m.horizonator; "
            "assert 'render' in dir(t); assert 'SRTM' in t.__doc__ and 'az_deg0' in t.render.__doc__; print('ok')" % outdir)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


STUBS = os.path.join(ROOT, "tests", "caller_stubs")


@pytest.fixture(scope="module")
def standalone(tmp_path_factory):
    """the reference's CLI: standalone.c + annotator.c compiled where they lie (BASELINE north_star: "so
    standalone.c links unchanged"; reference Makefile:21 puts annotator.c beside the library) against include/,
    with stand-ins for the CLI's OWN dependencies only (tests/caller_stubs: FreeImage, epoxy, freeglut, cairo,
    swscale - none in this image); every horizonator_* call resolves to libhorizonator.so"""
    out = tmp_path_factory.mktemp("standalone")
    exe = str(out / "standalone")
    cmd = ["gcc", "-std=gnu99", "-O1", "-w", "-I" + STUBS, "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "horizonator_amd", "csrc"),
           os.path.join(REF, "standalone.c"), os.path.join(REF, "annotator.c"), os.path.join(STUBS, "stubs.c"),
           "-o", exe, "-L" + LIBDIR, "-lhorizonator", "-Wl,-rpath," + LIBDIR, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "the reference's standalone.c / annotator.c no longer build and link against include/ + libhorizonator.so:\n" + r.stderr
    return exe


def test_standalone_and_annotator_link_unchanged(standalone):
    undefined = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--undefined-only", standalone], text=True).splitlines()}
    ours = {u for u in undefined if u.startswith("horizonator")}
    exported = {l.split()[-1] for l in subprocess.check_output(
        ["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libhorizonator.so")], text=True).splitlines()}
    # reference standalone.c:433-460 (init, set_zextents, pan_zoom, render_offscreen), :59-108 (redraw, resized: its window
    # mode), annotator.c:228-348 (unproject, project, x_from_az)
    assert ours == {"horizonator_init", "horizonator_set_zextents", "horizonator_pan_zoom", "horizonator_render_offscreen",
                    "horizonator_redraw", "horizonator_resized", "horizonator_project", "horizonator_unproject",
                    "horizonator_x_from_az"}, ours
    assert ours <= exported
    # annotate() is the reference's own annotator.c, compiled into the CLI: the library does not have to export it
    defined = subprocess.check_output(["nm", "--defined-only", standalone], text=True)
    assert " T annotate" in defined
    # nothing of the stand-ins leaks into the render path: they define no horizonator_* symbol
    # (no definition of, and no call to, anything of the API; its comments may name the library)
    import re
    stubs_text = open(os.path.join(STUBS, "stubs.c")).read()
    assert not re.search(r"\bhorizonator_\w+\s*\(", stubs_text)


def test_the_reference_cli_runs_up_to_the_device(standalone, tmp_path):
    """--help works; a render goes through horizonator_init() of THIS library, which without a HIP device (this
    container) fails with a message instead of falling back to anything"""
    r = subprocess.run([standalone, "--help"], capture_output=True, text=True)
    assert "--dirdems DIRECTORY" in r.stdout
    import hzutil
    if hzutil.hip_available():
        pytest.skip("a GPU is here: tests/test_gpu_standalone.py runs the CLI for real")
    dems = hzutil.dem_dir_for(hzutil.VIEW_LAT, hzutil.VIEW_LON, 200)
    r = subprocess.run([standalone, "--width", "400", "--height", "100", "--image", str(tmp_path / "out.png"), "--dirdems", dems,
                        "--zfar", "15000", str(hzutil.VIEW_LAT), str(hzutil.VIEW_LON), "0", "90"], capture_output=True, text=True)
    assert not (tmp_path / "out.png").exists()
    assert "horizonator_init() failed" in r.stderr and ("HIP" in r.stderr or "device" in r.stderr), r.stderr[-800:]
