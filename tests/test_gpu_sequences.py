"""Seeded random sequences of the API's calls on ONE context: renders into host memory, into
device memory, with and without the index map, sectors that change, picks (which read the
framebuffer after the conversion has cleared it: the draw is repeated), draws that are never
converted, packed and sparse strips - interleaved in random order, with one or two rounds per
draw forced at random; the viewer moves or only turns (draws from the vertex cache between cold ones), options change.  What the streams, events, three framebuffers and their flags have to
get right is the ORDER of things; every result is compared with the oracle's render of the view
that was current when it was asked for."""
import numpy as np
import pytest

import hzutil
import oracle

pytestmark = pytest.mark.gpu
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON


@pytest.mark.parametrize("seed", range(24))
def test_random_call_sequences_against_the_oracle(seed, monkeypatch):
    import torch
    import horizonator_amd
    from horizonator_amd.sharding import sparse_header_words, sparse_mask_stride
    rng = np.random.default_rng(4242 + seed)
    monkeypatch.setenv("HZ_TWO_PASS", str(int(rng.integers(0, 2))))
    R = int(rng.choice([120, 200, 300]))
    W = int(rng.choice([640, 1000, 1201]))          # 1201: no 4-pixel conversion kernel
    H = int(rng.choice([160, 251]))
    d = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, d, radius_cells=R)
    m = od.mosaic()
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=d, render_radius_cells=R)
    dev = torch.device("cuda:0")
    cache = {}

    def new_view():
        span = float(rng.choice([360.0, rng.uniform(20.0, 300.0)]))
        a0 = float(rng.uniform(-180.0, 180.0 - min(span, 359.0))) if span < 360.0 else -180.0
        frac = R / 1200.0 * 0.5
        return dict(lat=LAT + float(rng.uniform(-frac, frac)), lon=LON + float(rng.uniform(-frac, frac)),
                    a0=a0, a1=a0 + span, zfar=float(rng.choice([6000.0, 20000.0, 90000.0])))

    def want(view, c0, c1):
        key = (tuple(sorted(view.items())), c0, c1)
        if key not in cache:
            v = od.view(view["lat"], view["lon"], W, H, view["a0"], view["a1"], zfar=view["zfar"])
            cache[key] = oracle.render(m, v, W, H, c0, c1)
        return cache[key]

    try:
        view, c0, c1 = new_view(), 0, W
        h.set_view(view["a0"], view["a1"], lat=view["lat"], lon=view["lon"], zfar=view["zfar"])
        for step in range(40):
            op = rng.choice(["view", "turn", "turn", "options", "sector", "host", "full", "device", "device_ranges_only", "pick", "draw_only", "packed", "sparse"])
            what = f"seed {seed} step {step} {op} sector [{c0},{c1}) {view}"
            if op == "view":
                view = new_view()
                h.set_view(view["a0"], view["a1"], lat=view["lat"], lon=view["lon"], zfar=view["zfar"])
            elif op == "turn":
                # the viewer stays where it is and looks elsewhere (other azimuths, another far clip): from the second draw from
                # a viewpoint on, the view-independent half of the transform comes from the vertex cache (hz_draw.cpp)
                t = new_view()
                view = dict(view, a0=t["a0"], a1=t["a1"], zfar=t["zfar"])
                h.set_view(view["a0"], view["a1"], zfar=view["zfar"])
            elif op == "options":
                # none of them may change a byte: the vertex cache off / on (a cold draw between cached ones), host results in
                # 1..3 sectors
                h.set_options(vertex_cache=int(rng.integers(0, 2)), host_sectors=int(rng.integers(0, 4)))
            elif op == "sector":
                c0 = int(rng.integers(0, W - 8)) if rng.integers(0, 3) else 0
                c1 = int(rng.integers(c0 + 4, W + 1)) if rng.integers(0, 3) else W
                h.set_sector(c0, c1)
            elif op == "host":
                img, rngs = h.render(view["a0"], view["a1"], lat=view["lat"], lon=view["lon"], zfar=view["zfar"])
                o = want(view, c0, c1)
                assert np.array_equal(img, o["bgr"]) and np.array_equal(rngs, o["ranges"]), what
            elif op == "full":
                img, rngs, index, z24 = h.render_full(view["a0"], view["a1"], lat=view["lat"], lon=view["lon"], zfar=view["zfar"])
                hzutil.assert_same_render(dict(bgr=img, ranges=rngs, index=index, z24=z24), want(view, c0, c1), what)
            elif op in ("device", "device_ranges_only", "draw_only"):
                SW = c1 - c0
                d_img = torch.empty((H, SW, 3), dtype=torch.uint8, device=dev)
                d_rng = torch.empty((H, SW), dtype=torch.float32, device=dev)
                if op == "draw_only":
                    h.render_device(0, 0)                       # a draw nobody converts
                    continue
                h.render_device(d_img.data_ptr() if op == "device" else 0, d_rng.data_ptr())
                h.sync()
                o = want(view, c0, c1)
                assert np.array_equal(d_rng.cpu().numpy(), o["ranges"]), what
                if op == "device":
                    assert np.array_equal(d_img.cpu().numpy(), o["bgr"]), what
            elif op == "pick":
                o = want(view, c0, c1)
                h.render_device(0, 0)
                terrain = np.argwhere(o["index"] >= 0)
                sky = np.argwhere(o["index"] < 0)
                if len(terrain):
                    y, x = terrain[int(rng.integers(0, len(terrain)))]
                    assert h.pick(int(x) + c0, int(y)) is not None, what
                if len(sky):
                    y, x = sky[int(rng.integers(0, len(sky)))]
                    assert h.pick(int(x) + c0, int(y)) is None, what
            elif op in ("packed", "sparse"):
                SW = c1 - c0
                d_img = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
                d_rng = torch.zeros((H, W), dtype=torch.float32, device=dev)
                if op == "packed":
                    d_pk = torch.empty((H, SW), dtype=torch.int32, device=dev)
                    h.render_packed(d_pk.data_ptr())
                    h.resolve_packed(d_pk.data_ptr(), SW, SW, c0, d_img.data_ptr(), d_rng.data_ptr())
                else:
                    ms = sparse_mask_stride(SW)
                    d_sp = torch.zeros(sparse_header_words(H, ms) + H * SW, dtype=torch.int32, device=dev)
                    h.render_sparse(d_sp.data_ptr(), ms)
                    h.resolve_sparse_gathered([(d_sp.data_ptr(), c0, SW)], ms, d_img.data_ptr(), d_rng.data_ptr())
                h.sync()
                o = want(view, c0, c1)
                assert np.array_equal(d_img[:, c0:c1].cpu().numpy(), o["bgr"]), what
                assert np.array_equal(d_rng[:, c0:c1].cpu().numpy(), o["ranges"]), what
    finally:
        h.close()
