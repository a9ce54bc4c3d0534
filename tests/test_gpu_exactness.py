"""The two shortcuts of the marching kernel that rest on an argument instead of on the reference's
arithmetic, checked on the device against that arithmetic on seeded inputs around every border of
the argument ("edge-on slivers at x ~ 32768 with 10^5 LSB/px gradients are exactly
where a scene does not go looking").

* hz_tri_hidden() (hz_raster.h): the second round's early depth test skips a triangle when every
  pixel centre of its box holds a nearer depth than any fragment of the triangle can have.  GL's
  rule being protected: depth test GL_LESS (reference horizonator-lib.c:183-185) - a skipped
  triangle must not have been able to win a pixel.  2^30 seeded triangles per image size; for each,
  the LARGEST stored depth that still reads "hidden" is found by bisection and every covered pixel
  centre is drawn with the rasteriser's own planes: no fragment depth may be <= that stored depth.
* mr_simple_cull() (hz_k_march.h): rows of cells whose vertices all lie inside the view volume
  are culled with a back-face test and a pixel box alone.  Rules being protected: reference
  geometry.glsl:21-27 (triangles wider than a quarter of the image are dropped), GL's cull and
  scissor - its verdict must be hz_tri_cull()'s on every cell of every row it accepts.
* hiz_rect_min_depth() (hz_k_hiz.h): in zoomed views k_big drops a chunk of a large triangle's rows when the smallest
  depth any fragment of the chunk's rectangle can get lies behind everything already drawn there; that smallest depth
  is taken from the rectangle's four corners (every step of hz_tri_fragment's depth is monotone in px and py).
  Same GL rule; checked against the minimum over every pixel centre of seeded rectangles and depth planes.
"""
import ctypes as C

import pytest

from horizonator_amd import _lib as hzlib

pytestmark = pytest.mark.gpu


def _run(what, seed, n, W, H, col0=0, col1=None):
    lib = hzlib.load_selftest()          # (libhorizonator_selftest.so: the library's sources + the check kernels, include/hz_selftest.h)
    out = (C.c_uint64 * 5)()
    rc = lib.hz_hip_check_exactness(0, what, seed, n, W, H, col0, W if col1 is None else col1, out)
    assert rc == 0, lib.hz_hip_last_error()
    return [int(x) for x in out]


@pytest.mark.parametrize("W,H", [(16000, 4000), (32768, 8192), (2000, 500)])
def test_no_triangle_the_early_depth_test_skips_could_have_won_a_pixel(W, H):
    tested, hidden, frags, violations, margin = _run(0, 0x5EED0000 + W, 1 << 30, W, H)
    assert violations == 0, f"{violations} fragments at or in front of the depth their triangle was 'hidden' behind"
    # the check must have looked where it claims to: most triangles survive the cull in one winding or the other,
    # most of those are hidden behind SOME depth, and they cover pixels
    assert tested > (1 << 30) // 3 and hidden > tested // 2 and frags > hidden // 4, (tested, hidden, frags)
    assert margin - (1 << 32) >= 1, margin          # every fragment strictly behind the largest 'hidden' depth
    print(f"{W}x{H}: {tested} triangles, {hidden} hidden behind some depth, {frags} fragments drawn, "
          f"closest call {margin - (1 << 32)} LSB")


@pytest.mark.parametrize("W,H,col0,col1", [(16000, 4000, 0, 16000), (16000, 4000, 6000, 8000), (32768, 8192, 0, 32768),
                                           (1000, 250, 0, 1000), (65535, 4000, 100, 65000)])
def test_the_cull_of_whole_cells_is_the_cull_of_each_triangle(W, H, col0, col1):
    cases, shortway, cells, bad, kept = _run(1, 0xC0FFEE00 + W + col0, 1 << 22, W, H, col0, col1)
    assert bad == 0, f"{bad} of {2 * cells} triangle verdicts differ from hz_tri_cull()"
    assert cases == 1 << 22 and shortway > cases // 8 and cells == 63 * shortway, (cases, shortway, cells)
    assert 0 < kept < 2 * cells                     # both verdicts occur
    print(f"{W}x{H} columns [{col0},{col1}): {shortway} of {cases} row pairs culled the short way, {2 * cells} verdicts, {kept} kept")


@pytest.mark.parametrize("W,H", [(16000, 4000), (32768, 8192), (48, 48)])
def test_the_smallest_depth_of_a_rectangle_is_at_a_corner(W, H):
    n = 1 << 22
    cases, numbers, pixels, bad, nans = _run(2, 0xD0C70000 + W, n, W, H)
    assert bad == 0, f"{bad} rectangles whose smallest depth is not their corners' (or pixels hz_tri_fragment computes differently)"
    # the check must have looked where it claims to: most planes are numbers, some are not, and the rectangles have an inside
    assert cases == n and numbers > n // 2 and nans > n // 64 and pixels > 200 * n, (cases, numbers, pixels, nans)
    print(f"{W}x{H}: {cases} rectangles, {pixels} pixel centres, {nans} with a depth that is not a number")
