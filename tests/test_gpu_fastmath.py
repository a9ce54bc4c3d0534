"""horizonator_amd/csrc/hz_fast.h: the marching kernel's transform spells its IEEE divisions and
square roots as hipcc's own instruction sequences minus the parts that only act near the ends of
the float32 range.  Inside the operand range the kernel guarantees before using them, they must
be the device's `/` and sqrtf bit for bit: checked here on the device itself - exhaustively
where the operand is one float, on 2^32 seeded pairs for the division - and on the render as a
whole (abridged transform vs unabridged, every output)."""
import ctypes as C

import numpy as np
import pytest

import hzutil
from horizonator_amd import _lib as hzlib

pytestmark = pytest.mark.gpu


def _check(what, seed=0, n=0):
    lib = hzlib.load_selftest()          # (libhorizonator_selftest.so: the library's sources + the check kernels, include/hz_selftest.h)
    bad = C.c_uint64(12345)
    first = (C.c_float * 4)()
    assert lib.hz_hip_check_fastmath(0, what, seed, n, C.byref(bad), first) == 0, lib.hz_hip_last_error()
    assert bad.value == 0, f"{bad.value} mismatches, first: a={first[0]!r} b={first[1]!r} want={first[2]!r} got={first[3]!r}"


def test_reciprocal_every_float_in_range():
    """v_rcp_f32 + one Newton step is the correctly rounded reciprocal of every float with 2^-62 <= |b| <= 2^62 (round 5:
    tools/exact_seq.hip found it so on the MI355X; hipcc's own sequence has two more steps)"""
    _check(0)


def test_quotient_by_two_pi_every_numerator():
    """... and one correction step the correctly rounded a/(2 pi) of every numerator that is zero or in 2^-100 .. 2^30"""
    _check(6)


def test_min_over_max_with_one_is_the_number_or_its_reciprocal():
    """the arc tangent's min(t,1)/max(t,1): every t that is zero or in 2^-62 .. 2^62"""
    _check(7)


@pytest.mark.parametrize("what,seed", [(8, 21), (8, 22), (9, 23), (9, 24)])
def test_abridged_arc_tangent_seeded_pairs(what, seed):
    """hzf_atan2 (reciprocal given or not, the sign taken from the operands' sign bits) against hz_atan2: 2^30 pairs per seed"""
    _check(what, seed=seed * 0x3000000000, n=1 << 30)


def test_square_root_every_float_from_2_pow_minus_96():
    """round 5: v_rsq_f32 and one Newton step (five instructions instead of nine)"""
    _check(1)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_division_seeded_pairs(seed):
    _check(2, seed=seed * 0x1000000000, n=1 << 30)


@pytest.mark.parametrize("divisor", [6.28318548, 599900.0, 39900.0, 29900.0, 3.0, 1e-3, 12345.678])
def test_division_by_a_draw_constant_every_numerator(divisor):
    """2*pi and depth/colour extents like the benchmarks'"""
    bits = int(np.float32(divisor).view(np.uint32))
    _check(3, seed=bits)


def test_double_reciprocal_of_every_divisor_is_within_2_pow_minus_50():
    """hz_rcp_f64 (k_big's span division): v_rcp_f64 + two Newton steps, every d in [1, 2^31)"""
    _check(4)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_floor_division_against_64_bit_integer_division(seed):
    """hz_floor_div (the exact first/last covered column of a row): 2^30 seeded (n, d) per seed, all magnitudes,
    both signs, multiples of d and their neighbours - against the device's own int64 division"""
    _check(5, seed=seed * 0x2000000000, n=1 << 30)


def test_abridged_transform_renders_the_same_bytes(monkeypatch):
    import oracle
    LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON
    R, W, H = 500, 3000, 750
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    for kw in (dict(zfar=200000.0), dict(zfar=30000.0, znear=1.0, znear_color=50.0, zfar_color=9000.0), dict(viewer_z=4000.0, zfar=90000.0)):
        v = od.view(LAT, LON, W, H, -180, 180, **kw)
        monkeypatch.setenv("HZ_NO_FAST_MATH", "1")
        plain = hzutil.hip_render(m, v, W, H, raster=2)
        monkeypatch.delenv("HZ_NO_FAST_MATH")
        fast = hzutil.hip_render(m, v, W, H, raster=2)
        hzutil.assert_same_render(fast, plain, f"abridged vs unabridged transform {kw}")
        hzutil.assert_same_render(fast, oracle.render(m, v, W, H), f"abridged transform vs oracle {kw}")
