"""TEST INFRASTRUCTURE: regenerate tests/golden/*.npz from the reference itself.

Runs only where /root/reference and Mesa's llvmpipe exist (the build
container).  Two sources, both the reference's own code executed in place:
  dem_*.npz     reference dem.c compiled as oracle/_ref/libdem_ref.so
  vertex_*.npz  reference vertex.glsl on llvmpipe, transform-feedback capture
  render_*.npz  reference vertex/geometry/fragment.glsl on llvmpipe, full draw
The inputs of every fixture (DEM window, uniform values) are stored with the
outputs, so the tests need neither the reference nor the tile generator.
Uniform VALUES come from oracle.Dem.view(), i.e. from the restatement of the
reference's host code (reference horizonator-lib.c:765-799): that host code
cannot be built here, see oracle.h.

    python -m oracle.make_golden
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hzutil  # noqa: E402
import oracle  # noqa: E402
from oracle import glsl_run  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON


def view_arrays(v):
    return {"u_" + k: np.float32(val) for k, val in v.as_dict().items()}


def dem_fixtures():
    ref = oracle.load_ref_dem()
    cases = [  # lat, lon, radius_cells, radius_m, srtm1
        (LAT, LON, 32, -1.0, False),
        (LAT, LON, 600, -1.0, False),
        (LAT, LON, 1801, -1.0, False),
        (LAT, LON, -1, 40000.0, False),
        (33.9, -118.2, 777, -1.0, False),
        (LAT, LON, 300, -1.0, True),
    ]
    out = {}
    for k, (lat, lon, R, Rm, srtm1) in enumerate(cases):
        rc = 3000 if R < 0 else R
        d = hzutil.dem_dir_for(lat, lon, rc, srtm1=srtm1)
        ctx = oracle._RefDemCtx()
        ok = ref.horizonator_dem_init(C.byref(ctx), lat, lon, R, Rm, d.encode(), srtm1)
        assert ok, (lat, lon, R)
        N = 2 * ctx.radius_cells
        rng = np.random.default_rng(100 + k)
        # random samples plus the window's borders and the tile seams
        ii = np.concatenate([rng.integers(-2, N + 2, 4000), np.arange(-1, N + 1), np.arange(-1, N + 1)])
        jj = np.concatenate([rng.integers(-2, N + 2, 4000), np.full(N + 2, 0), np.full(N + 2, N - 1)])
        cpd = ctx.cells_per_deg
        seam = cpd - ctx.origin_dem_cellij[0]
        if 0 < seam < N:
            ii = np.concatenate([ii, np.full(50, seam), np.full(50, seam - 1), np.full(50, seam + 1)])
            jj = np.concatenate([jj, rng.integers(0, N, 150)])
        vals = np.array([ref.horizonator_dem_sample(C.byref(ctx), int(i), int(j)) for i, j in zip(ii, jj)], np.int16)
        b = [C.c_float() for _ in range(4)]
        ref.horizonator_dem_bounds_latlon_deg(C.byref(ctx), *[C.byref(x) for x in b])
        out[f"c{k}_args"] = np.array([lat, lon, R, Rm, int(srtm1)], np.float64)
        out[f"c{k}_window"] = np.array(list(ctx.origin_dem_lon_lat) + list(ctx.origin_dem_cellij) +
                                       list(ctx.Ndems_ij) + [ctx.radius_cells, ctx.cells_per_deg], np.int32)
        out[f"c{k}_bounds"] = np.array([x.value for x in b], np.float32)
        out[f"c{k}_ij"] = np.stack([ii, jj]).astype(np.int32)
        out[f"c{k}_z"] = vals
        ref.horizonator_dem_deinit(C.byref(ctx))
    out["ncases"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(OUT, "dem_samples.npz"), **out)
    print("dem_samples.npz:", len(cases), "cases")


def vertex_fixture(name, R, W, H, az0, az1, lat=LAT, lon=LON, **kw):
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    v = od.view(lat, lon, W, H, az0, az1, **kw)
    gv = glsl_run.vertices(m, v)
    np.savez_compressed(os.path.join(OUT, f"vertex_{name}.npz"), mosaic=m, W=np.int32(W), H=np.int32(H),
                        gl_position=gv[:, :, :4], red=gv[:, :, 4], **view_arrays(v))
    print(f"vertex_{name}.npz: {m.shape[0]}^2 vertices")


def render_fixture(name, R, W, H, az0, az1, lat=LAT, lon=LON, keep_depth=True, rough=False, **kw):
    d = hzutil.dem_dir_for(LAT, LON, R, rough=rough)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    v = od.view(lat, lon, W, H, az0, az1, **kw)
    g = glsl_run.render(m, v, W, H)
    extra = {"depth": g["depth"]} if keep_depth else {}
    np.savez_compressed(os.path.join(OUT, f"render_{name}.npz"), mosaic=m, W=np.int32(W), H=np.int32(H),
                        bgr=g["bgr"], z24=g["z24"], **extra, **view_arrays(v))
    print(f"render_{name}.npz: {W}x{H}, terrain fraction {(g['z24'] != 0xFFFFFF).mean():.3f}")


def checksum_fixtures():
    """full-size scenes, too large to commit as images: the SHA-256 of what the
    reference's shaders drew on llvmpipe (BGR bytes, 24-bit depth), next to the
    hash of the DEM window and the uniform values that went in"""
    import hashlib
    import json
    out = {}
    # name, R, W, H, azimuth extents, viewpoint offset from the window's centre (degrees), view keywords
    for name, R, W, H, az0, az1, dlat, dlon, kw in [
            ("cfg2_3x3_8000x2000", 1800, 8000, 2000, -180.0, 180.0, 0.0, 0.0, dict(zfar=600000.0)),
            ("cfg3_7x7_16000x4000", 4200, 16000, 4000, -180.0, 180.0, 0.0, 0.0, dict(zfar=600000.0)),
            ("cfg3_7x7_16000x4000_zfar40km", 4200, 16000, 4000, -180.0, 180.0, 0.0, 0.0, dict(zfar=40000.0)),
            # the benchmark's image and mosaic with other views: a 45 degree zoom, a viewer 4.5 km up, a moved viewpoint
            ("cfg3_7x7_16000x4000_zoom45deg", 4200, 16000, 4000, 30.0, 75.0, 0.0, 0.0, dict(zfar=150000.0)),
            ("cfg3_7x7_16000x4000_viewer_4500m", 4200, 16000, 4000, -180.0, 180.0, 0.0, 0.0, dict(zfar=600000.0, viewer_z=4500.0)),
            ("cfg2_3x3_8000x2000_moved_wide", 1800, 8000, 2000, -100.0, 140.0, 0.3, -0.2, dict(znear=10.0, zfar=200000.0))]:
        d = hzutil.dem_dir_for(LAT, LON, R)
        od = oracle.Dem(LAT, LON, d, radius_cells=R)
        m = od.mosaic()
        v = od.view(LAT + dlat, LON + dlon, W, H, az0, az1, **kw)
        g = glsl_run.render(m, v, W, H)
        out[name] = {"R": R, "W": W, "H": H, "lat": LAT, "lon": LON, "az_deg0": az0, "az_deg1": az1,
                     "view_lat": LAT + dlat, "view_lon": LON + dlon, "kw": kw,
                     "znear": float(kw.get("znear", 100.0)), "zfar": float(kw["zfar"]),
                     "view": {k: float(np.float32(x)) for k, x in v.as_dict().items()},
                     "mosaic_sha256": hashlib.sha256(m.tobytes()).hexdigest(),
                     "bgr_sha256": hashlib.sha256(g["bgr"].tobytes()).hexdigest(),
                     "z24_sha256": hashlib.sha256(g["z24"].tobytes()).hexdigest(),
                     "terrain_fraction": float((g["z24"] != 0xFFFFFF).mean())}
        print(f"checksum {name}: terrain fraction {out[name]['terrain_fraction']:.3f}", flush=True)
    json.dump(out, open(os.path.join(OUT, "render_checksums.json"), "w"), indent=1)


def batch_checksum_fixtures():
    """BASELINE.json configs[3]: viewpoints of the 16x16 lattice over one 5x5-tile
    window, 8000x2000 each - the SHA-256 of the reference's draw for a few of them"""
    import hashlib
    import json
    R, W, H, zfar = 3000, 8000, 2000, 600000.0
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    lats, lons = hzutil.viewpoint_lattice(LAT, LON)
    out = {"R": R, "W": W, "H": H, "lat": LAT, "lon": LON, "az_deg0": -180.0, "az_deg1": 180.0,
           "znear": 100.0, "zfar": zfar, "mosaic_sha256": hashlib.sha256(m.tobytes()).hexdigest(),
           "viewpoints": {}}
    for vp in (0, 15, 37, 68, 85, 102, 119, 136, 153, 170, 187, 204, 221, 240, 250, 255):
        v = od.view(float(lats[vp]), float(lons[vp]), W, H, -180.0, 180.0, zfar=zfar)
        g = glsl_run.render(m, v, W, H)
        out["viewpoints"][str(vp)] = {
            "lat": float(lats[vp]), "lon": float(lons[vp]),
            "view": {k: float(np.float32(x)) for k, x in v.as_dict().items()},
            "bgr_sha256": hashlib.sha256(g["bgr"].tobytes()).hexdigest(),
            "z24_sha256": hashlib.sha256(g["z24"].tobytes()).hexdigest(),
            "terrain_fraction": float((g["z24"] != 0xFFFFFF).mean())}
        print(f"batch checksum viewpoint {vp}: terrain fraction {out['viewpoints'][str(vp)]['terrain_fraction']:.3f}")
    json.dump(out, open(os.path.join(OUT, "batch_checksums.json"), "w"), indent=1)


def texture_fixtures():
    """the texture path ("next" row N4) on llvmpipe: texture coordinates of every vertex
    (transform feedback), raw textured triangles through the reference's fragment shader
    (sampler + blend, incl. wrap and non-power-of-two sizes), and whole textured draws.
    Textures are hzutil.hash_texture(): only their size and seed are stored."""
    # vertex stage
    R, W, H = 48, 256, 64
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    lat = LAT + 0.011
    v = od.view(lat, LON, W, H, -180, 180)
    t = od.texture(LAT, LON, viewer_lat=lat)
    cap = glsl_run.vertices_textured(m, v, t.as_glsl_job())
    np.savez_compressed(os.path.join(OUT, "tex_vertex.npz"), mosaic=m, N=m.shape[0], tex_st=cap[..., 5:7],
                        xyzr=cap[..., [0, 1, 2, 4]], init_lat=LAT, init_lon=LON, viewer_lat=lat,
                        **{"t_" + k: val for k, val in t.as_glsl_job().items()}, **view_arrays(v))
    print("tex_vertex.npz:", cap.shape)

    # sampler + blend probes: two triangles per draw, texture coordinates partly outside [0,1]
    rng = np.random.default_rng(2024)
    PW, PH = 96, 64
    draws = {}
    for k, (th, tw, blocky) in enumerate([(4, 4, 1), (80, 96, 1), (32, 64, 1), (5, 3, 1), (256, 768, 8), (768, 256, 1)]):
        s0, s1 = np.sort(rng.uniform(-0.3, 1.3, 2)) if k % 2 else (rng.uniform(0, .2), rng.uniform(.8, 1))
        t0, t1 = np.sort(rng.uniform(-0.3, 1.3, 2)) if k % 2 else (rng.uniform(0, .2), rng.uniform(.8, 1))
        r = rng.uniform(0, 1, 4)
        ds, dt = rng.uniform(-.1, .1, 2)
        A = [-1, -1, 0, r[0], s0, t0]; B = [1, -1, 0, r[1], s1, t0 + dt]
        Cc = [1, 1, 0, r[2], s1 + ds, t1 + dt]; D = [-1, 1, 0, r[3], s0 + ds, t1]
        tris = np.array([[A, B, Cc], [A, Cc, D]], np.float32)
        g = glsl_run.textured_triangles(tris, PW, PH, hzutil.hash_texture(th, tw, seed=k, blocky=blocky))
        draws[f"tris{k}"] = tris; draws[f"tex{k}"] = np.array([th, tw, k, blocky]); draws[f"bgr{k}"] = g["bgr"]
    np.savez_compressed(os.path.join(OUT, "tex_probe.npz"), W=PW, H=PH, n=6, **draws)
    print("tex_probe.npz: 6 draws")

    # whole draws
    for name, R, W, H, az0, az1, dlat, dlon, blocky, kw in [
            ("T1_full360", 64, 512, 128, -180, 180, 0.0, 0.0, 1, dict(zfar=8000.0)),
            ("T2_clipped", 150, 800, 200, -180, 180, 0.0, 0.0, 8, dict(zfar=7000.0, viewer_z=2300.0)),
            ("T3_moved",   32, 400, 160, 30, 140, 0.006, -0.004, 4, dict(zfar=5000.0, znear_color=300.0, zfar_color=2500.0))]:
        d = hzutil.dem_dir_for(LAT, LON, R)
        od = oracle.Dem(LAT, LON, d, radius_cells=R)
        m = od.mosaic()
        v = od.view(LAT + dlat, LON + dlon, W, H, az0, az1, **kw)
        t = od.texture(LAT, LON, viewer_lat=LAT + dlat)
        texels = hzutil.hash_texture(t.tex_h, t.tex_w, seed=7, blocky=blocky)
        g = glsl_run.render_textured(m, v, W, H, t.as_glsl_job(), texels)
        np.savez_compressed(os.path.join(OUT, f"texrender_{name}.npz"), mosaic=m, N=m.shape[0], W=W, H=H,
                            bgr=g["bgr"], z24=g["z24"], tex_seed=7, tex_blocky=blocky, tex_h=t.tex_h, tex_w=t.tex_w,
                            init_lat=LAT, init_lon=LON, viewer_lat=LAT + dlat, viewer_lon=LON + dlon,
                            **{"t_" + k: val for k, val in t.as_glsl_job().items()}, **view_arrays(v))
        print(f"texrender_{name}.npz: {W}x{H}, texture {t.tex_w}x{t.tex_h}, terrain fraction {(g['z24'] != 0xFFFFFF).mean():.3f}")


def random_checksum_fixtures():
    """64 seeded random configurations (hzutil.random_view_case): SHA-256 of the reference's draw"""
    import hashlib
    import json
    out = {}
    for seed in range(64):
        c = hzutil.random_view_case(seed)
        d = hzutil.dem_dir_for(LAT, LON, c["R"], rough=c["rough"])
        od = oracle.Dem(LAT, LON, d, radius_cells=c["R"])
        m = od.mosaic()
        v = od.view(c["lat"], c["lon"], c["W"], c["H"], c["az0"], c["az1"], **c["kw"])
        g = glsl_run.render(m, v, c["W"], c["H"])
        out[str(seed)] = {"mosaic_sha256": hashlib.sha256(m.tobytes()).hexdigest(),
                          "bgr_sha256": hashlib.sha256(g["bgr"].tobytes()).hexdigest(),
                          "z24_sha256": hashlib.sha256(g["z24"].tobytes()).hexdigest(),
                          "terrain_fraction": float((g["z24"] != 0xFFFFFF).mean())}
        print(f"random {seed}: R={c['R']} {c['W']}x{c['H']} terrain {out[str(seed)]['terrain_fraction']:.3f}")
    json.dump(out, open(os.path.join(OUT, "random_checksums.json"), "w"), indent=1)


def raster_probe_fixture():
    """llvmpipe's fill rule and depth rounding on hand-made triangles (our own
    pass-through shaders; no reference code involved)"""
    W = H = 16
    tris = []

    def px2ndc(x, y):       # window coordinates -> clip space for a WxH viewport
        return x / (W / 2.0) - 1.0, y / (H / 2.0) - 1.0

    def tri(p0, p1, p2, z=0.0, r=0.5):
        tris.append([[*px2ndc(*p), z, r] for p in (p0, p1, p2)])

    # vertices exactly on pixel centres: which edges own their pixels?
    tri((2.5, 2.5), (6.5, 2.5), (2.5, 6.5), z=-0.5)        # ccw right triangle
    tri((8.5, 8.5), (12.5, 8.5), (12.5, 12.5), z=-0.25)    # ccw
    tri((8.5, 8.5), (12.5, 12.5), (8.5, 12.5), z=-0.25)    # shares the diagonal
    tri((2.5, 9.5), (2.5, 13.5), (6.5, 9.5), z=0.0)        # cw: culled
    tris_a = np.array(tris, np.float32)
    g = glsl_run.raw_triangles(tris_a, W, H)
    np.savez_compressed(os.path.join(OUT, "raster_probe.npz"), tris=tris_a, W=np.int32(W), H=np.int32(H),
                        bgr=g["bgr"], z24=g["z24"], depth=g["depth"])
    print("raster_probe.npz")


def main():
    if not glsl_run.available() or oracle.load_ref_dem() is None:
        sys.exit("needs /root/reference and oracle/_ref (run `make -C oracle` first)")
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["checksums"]:           # only tests/golden/render_checksums.json
        checksum_fixtures()
        return
    if sys.argv[1:] == ["batch"]:               # only tests/golden/batch_checksums.json
        batch_checksum_fixtures()
        return
    if sys.argv[1:] == ["random"]:              # only tests/golden/random_checksums.json
        random_checksum_fixtures()
        return
    if sys.argv[1:] == ["texture"]:             # only the texture path's fixtures
        texture_fixtures()
        return
    dem_fixtures()
    raster_probe_fixture()
    # vertex stage (pins reference vertex.glsl:111-162 bit for bit)
    vertex_fixture("full360", 64, 256, 64, -180, 180)
    vertex_fixture("partial", 64, 300, 100, -40, 100, znear_color=50.0, zfar_color=5000.0, zfar=6000.0)
    vertex_fixture("wrapped", 64, 1000, 300, -710, -625, zfar=3000.0)
    # full draws (G1..G5)
    render_fixture("G1_partial", 32, 256, 64, -40, 100)
    render_fixture("G2_full360", 32, 256, 64, -180, 180)
    render_fixture("G3_cfg1", 600, 2000, 500, -180, 180, keep_depth=False)
    for k, (dlat, dlon) in enumerate([(0.0, 0.0), (0.013, -0.021), (-0.02, 0.017), (0.031, 0.029)]):
        render_fixture(f"G4_move{k}", 64, 512, 128, -180, 180, lat=LAT + dlat, lon=LON + dlon, zfar=8000.0)
    render_fixture("G5_zextents", 64, 512, 128, 20, 200, znear=300.0, zfar=5000.0, znear_color=1000.0, zfar_color=2500.0)
    render_fixture("G6_rough", 150, 900, 240, -180, 180, zfar=60000.0, rough=True)
    render_fixture("G7_zoom", 200, 600, 450, 40, 52, zfar=30000.0)
    render_fixture("G8_on_vertex", 64, 512, 128, -180, 180, lat=34.0 + 500 / 1200.0, lon=-118.0 + 500 / 1200.0)
    checksum_fixtures()
    batch_checksum_fixtures()
    random_checksum_fixtures()
    texture_fixtures()


if __name__ == "__main__":
    main()
