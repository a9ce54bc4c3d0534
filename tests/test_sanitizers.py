"""the host code under sanitizers.  `make asan` builds libhorizonator_asan.so - hz_dem.c,
hz_host.c, hz_png.c and hz_scatter.c compiled with -fsanitize=address,undefined (no recovery), linked with the same kernel
object - and the DEM, PNG, malformed-input and blob-scatter tests run once more through it in a child interpreter
(LD_PRELOAD=libasan, HORIZONATOR_AMD_LIB pointing at the sanitized build).  CPU only: neither this
container nor the GPU pool offers a device-side sanitizer."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gcc_file(name):
    return subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()


@pytest.mark.skipif(not os.path.isabs(_gcc_file("libasan.so")), reason="gcc has no libasan here")
def test_host_code_under_address_and_ub_sanitizers():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "horizonator_amd", "csrc"), "asan"], stdout=sys.stderr)
    lib = os.path.join(ROOT, "horizonator_amd", "libhorizonator_asan.so")
    assert os.path.exists(lib)
    env = dict(os.environ,
               HORIZONATOR_AMD_LIB=lib,
               LD_PRELOAD=_gcc_file("libasan.so") + ":" + _gcc_file("libubsan.so"),
               # python and numpy "leak" by design; any memory error or undefined behaviour in our code aborts the child
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               HZ_UNDER_SANITIZER="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_dem.py"), os.path.join(ROOT, "tests", "test_png.py"),
                        os.path.join(ROOT, "tests", "test_malformed_inputs.py"),
                        os.path.join(ROOT, "tests", "test_scatter.py")],       # (round 4: the host half of results-without-the-sky, hz_scatter.c)
                       cwd=ROOT, env=env, capture_output=True, text=True)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, "tests failed under the sanitizers:\n" + tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
