"""The work lists of sector draws (horizonator_amd/csrc/hz_kernels.hip: strips_behind_columns).

A draw of an azimuth sector - one GPU's share of a panorama - or of a view of
less than 360 degrees launches one marching wave per (segment of rows, strip column) pair that
the HOST lists as able to reach the drawn columns, instead of the whole grid.  The list only
has to be a superset of the waves that draw anything (the kernel repeats the decision with
the rasteriser's own arithmetic), but it must be that: a missing wave is missing terrain.
Checked here without a GPU against a float64 restatement of the vertex stage's azimuth
(reference vertex.glsl:128-152): every patch with a vertex inside the columns is listed, and
the list is not much longer than that.  The GPU parity tests then compare whole sector renders
with the oracle."""
import ctypes as C

import numpy as np
import pytest

from horizonator_amd import _lib as hzlib

MR_COLS = 63


def _view(rng, N, W, H, az0, az1):
    v = hzlib.View()
    v.viewer_cell_i = float(rng.uniform(0.02, 0.98) * N)
    v.viewer_cell_j = float(rng.uniform(0.02, 0.98) * N)
    v.viewer_z = 1500.0
    v.cos_viewer_lat = float(np.cos(np.radians(rng.uniform(5.0, 65.0))))
    v.deg_per_cell = 1.0 / 1200.0
    v.az_deg0, v.az_deg1 = az0, az1
    v.aspect = W / H
    v.znear, v.zfar = 100.0, 600000.0
    v.znear_color, v.zfar_color = 100.0, 600000.0
    return v


def _pixel_x(v, N, W):
    """window x of every vertex of the grid, float64 (reference vertex.glsl:128-152)"""
    K = 6371000.0 * np.pi / 180.0 * v.deg_per_cell
    e = (np.arange(N) - float(v.viewer_cell_i)) * K * float(v.cos_viewer_lat)
    n = (np.arange(N) - float(v.viewer_cell_j)) * K
    az = np.arctan2(e[None, :], n[:, None])
    az0 = np.radians(float(v.az_deg0))
    d = ((np.radians(float(v.az_deg1)) - np.pi) - az0) / (2 * np.pi)
    span = 2 * np.pi * (d - np.rint(d)) + np.pi
    center = az0 + span / 2
    dd = (az - center) / (2 * np.pi)
    x = 2 * np.pi * (dd - np.rint(dd)) * (2.0 / span)
    return x * W / 2 + W / 2


def _listed(N, W, H, v, c0, c1, rnd):
    lib = hzlib.load_selftest()          # (hz_hip_debug_worklist is a diagnostics entry point: include/hz_selftest.h)
    cap = 1 << 20
    out = np.zeros((cap, 3), np.int32)
    n = lib.hz_hip_debug_worklist(N, W, H, C.byref(v), c0, c1, rnd, out.ctypes.data, cap)
    return n, out[:max(n, 0)]


CASES = [(seed, G) for seed in range(12) for G in (2, 3, 8)]


@pytest.mark.parametrize("seed,G", CASES)
def test_every_strip_that_reaches_the_sector_is_listed(seed, G):
    rng = np.random.default_rng(7000 + seed)
    N = int(rng.choice([400, 700, 1000]))
    W = int(rng.choice([1000, 4000, 16000]))
    H = W // 4
    if seed % 3 == 0:
        az0 = float(rng.uniform(-400, 400)); az1 = az0 + 360.0
    else:
        az0 = float(rng.uniform(-400, 400)); az1 = az0 + float(rng.uniform(5.0, 350.0))
    v = _view(rng, N, W, H, az0, az1)
    x = _pixel_x(v, N, W)
    edges = np.linspace(0, W, G + 1).astype(int)
    total_needed = total_listed = 0
    for g in range(G):
        c0, c1 = int(edges[g]), int(edges[g + 1])
        for rnd in (0, 1, 2):
            n, items = _listed(N, W, H, v, c0, c1, rnd)
            if n < 0:
                assert G == 1
                continue
            assert n == len(items)
            if rnd:
                continue                    # (the rounds partition round 0's segments differently: checked below)
            inside = (x >= c0 - 1.0) & (x <= c1 + 1.0)
            got = set()
            for sx, jb, je in items:
                got.add((int(sx), int(jb), int(je)))
            _, every = _listed(N, W, H, v, c0, c1, 256)        # every wave of the grid: the draw's segments
            segs = sorted({(int(jb), int(je)) for _, jb, je in every})
            assert got <= {tuple(int(t) for t in r) for r in every}
            nsx = (N - 1 + MR_COLS - 1) // MR_COLS
            needed = 0
            for jb, je in segs:
                rows = inside[jb:je + 1]
                for sx in range(nsx):
                    i0 = sx * MR_COLS
                    if rows[:, i0:min(i0 + MR_COLS, N - 1) + 1].any():
                        needed += 1
                        assert (sx, jb, je) in got, (seed, G, g, sx, jb, je)
            total_needed += needed
            total_listed += n
            assert n <= needed + 2 * len(segs) + 8, (seed, G, g, n, needed, len(segs))      # (round 5: 5 per segment)
    assert total_listed >= total_needed


def test_rounds_partition_the_one_round_list():
    """the two rounds of a two-round draw list disjoint waves that together cover the rows and
    columns of the one-round list (their segments are cut the same way)"""
    rng = np.random.default_rng(1)
    N, W, H = 1000, 16000, 4000
    v = _view(rng, N, W, H, -180.0, 180.0)
    c0, c1 = 3000, 5000
    n1, i1 = _listed(N, W, H, v, c0, c1, 1)
    n2, i2 = _listed(N, W, H, v, c0, c1, 2)
    s1 = {tuple(int(t) for t in r) for r in i1}
    s2 = {tuple(int(t) for t in r) for r in i2}
    assert n1 > 0 and n2 > 0 and not (s1 & s2)
    cover = np.zeros((N, (N - 1 + MR_COLS - 1) // MR_COLS), bool)
    for sx, jb, je in s1 | s2:
        assert not cover[jb:je, sx].any()           # no row of a strip column twice
        cover[jb:je, sx] = True
    n0, i0 = _listed(N, W, H, v, c0, c1, 0)
    for sx, jb, je in i0:
        # (round 0 cuts the far zones as the rounds do - same sector width - so its waves are covered)
        assert cover[jb:je, sx].all()


def test_full_circle_launches_the_grid():
    rng = np.random.default_rng(2)
    v = _view(rng, 600, 2000, 500, -180.0, 180.0)
    n, _ = _listed(600, 2000, 500, v, 0, 2000, 0)
    assert n == -1
    v2 = _view(rng, 600, 2000, 500, 10.0, 55.0)
    n, items = _listed(600, 2000, 500, v2, 0, 2000, 0)
    assert n > 0                                    # a 45 degree view: only the strips behind it
    nsx = (600 - 1 + MR_COLS - 1) // MR_COLS
    assert n < 0.6 * nsx * len({(int(jb), int(je)) for _, jb, je in items}) + 50
