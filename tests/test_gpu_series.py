"""The kernels that are TIMED, against the oracle.  bench.py queues K panoramas back to back; from the second on a draw
finds the marching kernel of the draw before it still running, and its second round then waits for its first, keeps
coarse depth (hz_k_hiz.h: k_march<false, true>, k_hiz, k_big dropping chunks of rows) and reads before its atomics -
a different instance of the marching kernel and different queue kernels than a single render that is waited for, which
is what the other parity tests draw.  Here: the LAST panorama of such a series, default switches, on every output
(visible-triangle index, 24-bit depth, BGR, ranges) against oracle/ and - where the reference can draw the scene -
against the SHA-256 of what its shaders drew on llvmpipe (tests/golden/render_checksums.json), at BASELINE's sizes:
configs[2] (the headline: 7x7 tiles, 16000x4000, far clip 600 km), the same with the API's 40 km far clip (reference
horizonator.h:10), and configs[4] (11x11 SRTM1 tiles, 32768x8192: the oracle on a 1/16 azimuth sector of the image).
The draw being reproduced: reference horizonator-lib.c:887-899, depth test :183-185."""
import hashlib
import json
import os

import numpy as np
import pytest

import hzutil
import oracle

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "render_checksums.json")))
LAT, LON = hzutil.VIEW_LAT, hzutil.VIEW_LON


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(autouse=True)
def _default_switches(monkeypatch):
    # (tools/gpu_modes.sh runs the suite under switches that change the plan: these tests are about what ships)
    for k in [k for k in os.environ if k.startswith("HZ_") and k != "HZ_TEST_DEM_DIR"]:
        monkeypatch.delenv(k)


def _last_of_a_series(h, W, H, n=8, coarse=True):
    """n panoramas queued back to back, nobody waits in between; returns the last one's four outputs (host arrays)
    and what the plan of that last draw was.  coarse: the draws of such a series keep coarse depth (hz_k_hiz.h) - not with a far
    clip so close that even the farthest cell is four pixels wide (the API's 40 km at 16000 columns: round 6)"""
    import torch
    dev = torch.device("cuda:0")
    out = {"bgr": torch.empty((H, W, 3), dtype=torch.uint8, device=dev), "ranges": torch.empty((H, W), dtype=torch.float32, device=dev),
           "index": torch.empty((H, W), dtype=torch.int32, device=dev), "z24": torch.empty((H, W), dtype=torch.int32, device=dev)}
    plan = None
    for attempt in range(3):        # (whether a draw finds its predecessor still marching is a race with the host; it practically always does)
        for _ in range(n):
            h.render_device(out["bgr"].data_ptr(), out["ranges"].data_ptr(), out["index"].data_ptr(), out["z24"].data_ptr())
        plan = h.last_plan()
        h.sync()
        if plan["coarse_depth"] or not coarse:
            break
    assert plan["rounds"] == 2 and bool(plan["coarse_depth"]) == coarse, f"the last of {n} renders queued back to back: coarse depth expected {coarse}: {plan}"
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("name", ["cfg3_7x7_16000x4000", "cfg3_7x7_16000x4000_zfar40km"])
def test_the_last_panorama_of_a_series_equals_oracle_and_reference(name):
    import horizonator_amd
    c = GOLD[name]
    R, W, H = c["R"], c["W"], c["H"]
    dems = hzutil.dem_dir_for(LAT, LON, R)
    od = oracle.Dem(LAT, LON, dems, radius_cells=R)
    m = od.mosaic()
    assert _sha(m) == c["mosaic_sha256"], "the synthetic DEM differs from the one the reference hashes were made on"
    v = od.view(LAT, LON, W, H, c["az_deg0"], c["az_deg1"], znear=c["znear"], zfar=c["zfar"])
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R)
    try:
        h.set_view(c["az_deg0"], c["az_deg1"], znear=c["znear"], zfar=c["zfar"])
        assert {k: np.float32(x) for k, x in v.as_dict().items()} == {k: np.float32(x) for k, x in h.view().items()}
        got = _last_of_a_series(h, W, H, coarse=not name.endswith("zfar40km"))
    finally:
        h.close()
    got["z24"] = got["z24"].view(np.uint32)
    orc = oracle.render(m, v, W, H)
    hzutil.assert_same_render(got, orc, name + ", last of a series")
    assert _sha(got["bgr"]) == c["bgr_sha256"] and _sha(got["z24"]) == c["z24_sha256"], "not the bytes the reference's shaders drew on llvmpipe"


def test_the_last_cfg5_panorama_of_a_series_equals_the_oracle_on_a_sixteenth():
    """BASELINE.json configs[4] (beyond what the reference can load: the oracle is the checker)"""
    import horizonator_amd
    R, W, H, zfar = 19800, 32768, 8192, 600000.0
    dems = hzutil.dem_dir_for(LAT, LON, R, srtm1=True)
    od = oracle.Dem(LAT, LON, dems, radius_cells=R, srtm1=True)
    v = od.view(LAT, LON, W, H, -180.0, 180.0, znear=100.0, zfar=zfar)
    h = horizonator_amd.horizonator(LAT, LON, W, H, dir_dems=dems, render_radius_cells=R, SRTM1=True)
    try:
        h.set_view(-180.0, 180.0, znear=100.0, zfar=zfar)
        m = h.mosaic()
        got = _last_of_a_series(h, W, H, n=6)
    finally:
        h.close()
    c0, c1 = 11 * W // 16, 12 * W // 16
    orc = oracle.render(m, v, W, H, c0, c1)
    sect = {k: np.ascontiguousarray(a[:, c0:c1]) for k, a in got.items()}
    sect["z24"] = sect["z24"].view(np.uint32)
    hzutil.assert_same_render(sect, orc, f"cfg5, columns [{c0},{c1}) of the last of a series")
