"""Texture path ("next" row N4): the oracle against what the reference's own shaders did
on Mesa llvmpipe (fixtures made by oracle/make_golden.py texture).  Textures are
hzutil.hash_texture(): the fixtures hold size and seed only."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import hzutil
import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TEXTURED = sorted(os.path.basename(p)[10:-4] for p in glob.glob(os.path.join(GOLD, "texrender_*.npz")))


def tex_of(g):
    """oracle.OrcTex from the t_* entries of a fixture"""
    t = oracle.OrcTex()
    for ours, theirs in (("viewer_lat_rad", "viewer_lat_rad"), ("origin_cell_lon_deg", "origin_cell_lon_deg"),
                         ("origin_cell_lat_deg", "origin_cell_lat_deg"), ("lon0", "texturemap_lon0"),
                         ("lon1", "texturemap_lon1"), ("dlat0", "texturemap_dlat0"), ("dlat1", "texturemap_dlat1"),
                         ("dlat2", "texturemap_dlat2")):
        setattr(t, ours, float(g["t_" + theirs]))
    t.ntiles_x, t.ntiles_y = int(g["t_NtilesX"]), int(g["t_NtilesY"])
    t.lowest_x, t.lowest_y = int(g["t_osmtile_lowestX"]), int(g["t_osmtile_lowestY"])
    t.tex_w, t.tex_h = t.ntiles_x * 256, t.ntiles_y * 256
    return t


def test_texture_coordinates_are_bit_identical_to_the_reference_shader():
    g = np.load(os.path.join(GOLD, "tex_vertex.npz"))
    t = tex_of(g)
    lib = oracle.load()
    N = int(g["N"])
    st = np.empty((N, N, 2), np.float32)
    buf = (C.c_float * 2)()
    for j in range(N):
        for i in range(N):
            lib.orc_vertex_tex(C.byref(t), float(g["u_deg_per_cell"]), i, j, C.byref(buf))
            st[j, i] = buf[:]
    assert np.array_equal(st.view(np.uint32), g["tex_st"].view(np.uint32))
    assert 0.0 < st.min() and st.max() < 1.0            # the DEM window lies inside the tile mosaic


def test_host_side_texture_parameters():
    """tile range, grid origin and Taylor coefficients (reference horizonator-lib.c:225-246,372-389,
    577-582,707-759,801): restated from the reference's host code, which cannot be built here;
    this pins the restatement against drift and checks it against the tile formula itself"""
    g = np.load(os.path.join(GOLD, "tex_vertex.npz"))
    d = hzutil.dem_dir_for(float(g["init_lat"]), float(g["init_lon"]), 48)
    od = oracle.Dem(float(g["init_lat"]), float(g["init_lon"]), d, radius_cells=48)
    t, want = od.texture(float(g["init_lat"]), float(g["init_lon"]), viewer_lat=float(g["viewer_lat"])), tex_of(g)
    for name, _ in oracle.OrcTex._fields_[:-1]:
        assert getattr(t, name) == getattr(want, name), name
    # slippy-map tile of a point, double precision: https://wiki.openstreetmap.org/wiki/Slippy_map_tilenames
    lat, lon = np.radians(float(g["init_lat"])), float(g["init_lon"])
    n = 2 ** 12
    x = int((lon + 180.0) / 360.0 * n)
    y = int((1.0 - np.arcsinh(np.tan(lat)) / np.pi) / 2.0 * n)
    assert t.lowest_x <= x < t.lowest_x + t.ntiles_x and t.lowest_y <= y < t.lowest_y + t.ntiles_y
    # the quadratic in dlat reproduces the exact tile y of the viewer's latitude at dlat = 0
    assert abs(t.dlat0 - (1.0 - np.arcsinh(np.tan(np.radians(float(g["viewer_lat"])))) / np.pi) / 2.0 * n) < 2e-3


def test_sampler_and_blend_on_probe_triangles():
    """GL_LINEAR/GL_REPEAT sampling of RGB8 textures (power-of-two and not, coordinates beyond
    [0,1]) and fragment.glsl's blend, drawn by llvmpipe through the reference's fragment shader"""
    g = np.load(os.path.join(GOLD, "tex_probe.npz"))
    W, H = int(g["W"]), int(g["H"])
    for k in range(int(g["n"])):
        th, tw, seed, blocky = (int(x) for x in g[f"tex{k}"])
        o = oracle.draw_triangles(g[f"tris{k}"], W, H, hzutil.hash_texture(th, tw, seed=seed, blocky=blocky))
        assert np.array_equal(o["bgr"], g[f"bgr{k}"]), (k, th, tw)


@pytest.mark.parametrize("name", TEXTURED)
def test_textured_draw_is_identical_to_the_reference_on_llvmpipe(name):
    g = np.load(os.path.join(GOLD, f"texrender_{name}.npz"))
    W, H = int(g["W"]), int(g["H"])
    v = oracle.make_view(**{k: float(g["u_" + k]) for k in oracle.VIEW_FIELDS})
    texels = hzutil.hash_texture(int(g["tex_h"]), int(g["tex_w"]), seed=int(g["tex_seed"]), blocky=int(g["tex_blocky"]))
    o = oracle.render(g["mosaic"], v, W, H, tex=tex_of(g), texels=texels, want=("bgr", "z24"))
    assert np.array_equal(o["z24"], g["z24"])
    assert np.array_equal(o["bgr"], g["bgr"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", TEXTURED)
def test_hip_textured_draw_is_identical_to_the_reference_on_llvmpipe(name):
    g = np.load(os.path.join(GOLD, f"texrender_{name}.npz"))
    W, H = int(g["W"]), int(g["H"])
    v = oracle.make_view(**{k: float(g["u_" + k]) for k in oracle.VIEW_FIELDS})
    texels = hzutil.hash_texture(int(g["tex_h"]), int(g["tex_w"]), seed=int(g["tex_seed"]), blocky=int(g["tex_blocky"]))
    hip = hzutil.hip_render(g["mosaic"], v, W, H, tex=tex_of(g), texels=texels)
    assert np.array_equal(hip["z24"], g["z24"])
    assert np.array_equal(hip["bgr"], g["bgr"])
