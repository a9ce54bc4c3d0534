"""hz_scatter.c (no GPU needed): the host half of "results into the caller's memory without the sky" - the sky's
constants (reference horizonator-lib.c:185, :1016) filled in by ranges of bytes, and blobs of terrain pixels
(format: hz_scatter.c; written on the device by k_pack_host) put in their places.  The blobs here come from a
numpy restatement of the format."""
import ctypes as C

import numpy as np
import pytest

from horizonator_amd import _lib as hzlib

ROWS, COLS = 4, 2048
RANGES, INDEX, Z24, RED = 1, 2, 4, 8


def _lib():
    lib = hzlib.load()
    lib.hz_sky_fill.restype = None
    lib.hz_sky_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
    lib.hz_blob_walk.restype = C.c_size_t
    lib.hz_blob_walk.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.hz_blob_scatter.restype = C.c_int
    lib.hz_blob_scatter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


SKY = {0: np.array([255, 0, 0], np.uint8).tobytes(), 1: np.float32(-1.0).tobytes(), 2: np.int32(-1).tobytes(), 3: np.uint32(0xFFFFFF).tobytes()}


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_sky_fill_writes_the_constant_into_exactly_the_bytes_asked_for(kind):
    lib = _lib()
    px = len(SKY[kind])
    rng = np.random.default_rng(kind)
    for trial in range(60):
        n = int(rng.integers(1, 700)) * px
        off = int(rng.integers(0, 16))                          # every alignment of the buffer itself
        raw = np.full(n + 64, 0x5A, np.uint8)
        buf = raw[off:off + n]
        lo = int(rng.integers(0, n)); hi = int(rng.integers(lo, n + 1))
        lib.hz_sky_fill(buf.ctypes.data, lo, hi, kind)
        want = np.full(n, 0x5A, np.uint8)
        pat = np.frombuffer(SKY[kind] * (n // px), np.uint8)
        want[lo:hi] = pat[lo:hi]
        assert np.array_equal(buf, want), (kind, n, off, lo, hi)
        assert (raw[:off] == 0x5A).all() and (raw[off + n:] == 0x5A).all()


def _blob(yo0, x0, n, terrain, rng, idx, z24, red, flags):
    """terrain: bool[4, n]; the value arrays [4, n]"""
    mw = (n + 31) // 32
    masks = np.zeros((ROWS, mw * 32), bool)
    masks[:, :n] = terrain
    words = np.packbits(masks.reshape(ROWS, mw, 32), axis=2, bitorder="little").view(np.uint32).reshape(ROWS, mw)
    T = terrain.sum(axis=1)
    body = [words.ravel()]
    if flags & RANGES: body.append(rng[terrain].view(np.uint32))
    if flags & INDEX:  body.append(idx[terrain].view(np.uint32))
    if flags & Z24:    body.append(z24[terrain])
    if flags & RED:
        r = red[terrain]
        r = np.concatenate([r, np.zeros((-len(r)) % 4, np.uint8)])
        body.append(r.view(np.uint32))
    body = np.concatenate(body).astype(np.uint32)
    size = 8 + len(body)
    pad = (-size) % 4
    hdr = np.array([yo0 | (flags << 16), x0, T[0], T[1], T[2], T[3], size + pad, n], np.uint32)
    return np.concatenate([hdr, body, np.zeros(pad, np.uint32)])


@pytest.mark.parametrize("flags", [RANGES | RED, RANGES | INDEX | Z24 | RED, RED, RANGES, INDEX | Z24])
def test_blobs_land_where_the_dense_copy_would_put_them(flags):
    lib = _lib()
    g = np.random.default_rng(flags)
    SW, H = 5000, 23                      # a last tile of 904 columns, a last blob of 3 rows
    terrain = g.random((H, SW)) < 0.6
    terrain[:, 100:400] = True            # whole mask words of terrain ...
    terrain[:, 2048:2048 + 64] = True
    terrain[4:8, :2048] = False           # ... and a tile without any: no blob
    rng = g.random((H, SW)).astype(np.float32) * 1e5
    idx = g.integers(0, 2**31 - 1, (H, SW)).astype(np.int32)
    z24 = g.integers(0, 0xFFFFFF, (H, SW)).astype(np.uint32)
    red = g.integers(0, 256, (H, SW)).astype(np.uint8)
    want = {"bgr": np.zeros((H, SW, 3), np.uint8), "ranges": np.where(terrain, rng, np.float32(-1)).astype(np.float32),
            "index": np.where(terrain, idx, -1).astype(np.int32), "z24": np.where(terrain, z24, 0xFFFFFF).astype(np.uint32)}
    want["bgr"][..., 0] = np.where(terrain, 0, 255); want["bgr"][..., 2] = np.where(terrain, red, 0)
    got = {"bgr": np.empty((H, SW, 3), np.uint8), "ranges": np.empty((H, SW), np.float32), "index": np.empty((H, SW), np.int32), "z24": np.empty((H, SW), np.uint32)}
    for kind, k in enumerate(("bgr", "ranges", "index", "z24")):
        lib.hz_sky_fill(got[k].ctypes.data, 0, got[k].nbytes, kind)
    blobs = []
    for yo0 in range(0, H, ROWS):
        for x0 in range(0, SW, COLS):
            n = min(COLS, SW - x0)
            def cut(a, fill):
                out = np.full((ROWS, n), fill, a.dtype)
                rows = min(ROWS, H - yo0)
                out[:rows] = a[yo0:yo0 + rows, x0:x0 + n]
                return out
            t = cut(terrain, False)
            if t.any():
                blobs.append(_blob(yo0, x0, n, t, cut(rng, 0), cut(idx, 0), cut(z24, 0), cut(red, 0), flags))
    assert len(blobs) < ((H + 3) // 4) * 3          # (the empty tile sent nothing)
    # a chunk: 12 words that a void of the chunk before reaches over, the blobs with a void of 8 words between the first two,
    # and a void at the end that reaches 20 words beyond the chunk
    VOID = 0xFFFFFFFE
    lead = np.array([7] * 12, np.uint32)
    gap = np.array([VOID, 8, 9, 9, 9, 9, 9, 9], np.uint32)
    tail = np.array([VOID, 24, 5, 5], np.uint32)
    chunk = np.concatenate([lead, blobs[0], gap] + blobs[1:] + [tail])
    offs = np.zeros(len(blobs) + 4, np.uint64)
    beyond = C.c_size_t(99)
    assert lib.hz_blob_walk(chunk.ctypes.data, len(chunk), 12, offs.ctypes.data, len(offs), C.byref(beyond)) == len(blobs)
    assert beyond.value == 20
    assert lib.hz_blob_walk(chunk.ctypes.data, len(chunk) - 9, 12, offs.ctypes.data, len(offs), C.byref(beyond)) == 2**64 - 1       # a cut-off blob is not a blob
    assert lib.hz_blob_walk(chunk.ctypes.data, len(chunk), 0, offs.ctypes.data, len(offs), C.byref(beyond)) == 2**64 - 1           # nor is rubbish
    assert lib.hz_blob_walk(chunk.ctypes.data, len(chunk), 12, offs.ctypes.data, len(offs), C.byref(beyond)) == len(blobs)
    at = 12
    for k, b in enumerate(blobs):
        if k == 1:
            at += 8
        assert int(offs[k]) == at
        rc = lib.hz_blob_scatter(chunk[at:].ctypes.data, SW, H,
                                 got["bgr"].ctypes.data if flags & RED else None, got["ranges"].ctypes.data if flags & RANGES else None,
                                 got["index"].ctypes.data if flags & INDEX else None, got["z24"].ctypes.data if flags & Z24 else None)
        assert rc == 0
        at += len(b)
    for k, f in (("bgr", RED), ("ranges", RANGES), ("index", INDEX), ("z24", Z24)):
        if flags & f:
            assert np.array_equal(got[k], want[k]), k
        else:                                       # a buffer the blobs do not carry stays sky
            sky = np.frombuffer(SKY[("bgr", "ranges", "index", "z24").index(k)] * (got[k].nbytes // len(SKY[("bgr", "ranges", "index", "z24").index(k)])), np.uint8)
            assert np.array_equal(got[k].view(np.uint8).ravel(), sky), k


def test_a_blob_that_is_not_one_is_refused():
    lib = _lib()
    t = np.ones((ROWS, 64), bool)
    z = np.zeros((ROWS, 64))
    b = _blob(0, 0, 64, t, z.astype(np.float32), z.astype(np.int32), z.astype(np.uint32), z.astype(np.uint8), RANGES | RED)
    out = np.zeros((8, 64), np.float32)
    img = np.zeros((8, 64, 3), np.uint8)
    assert lib.hz_blob_scatter(b.ctypes.data, 64, 8, img.ctypes.data, out.ctypes.data, None, None) == 0
    for field, value in ((1, 32), (7, 4096), (0, 8 | ((RANGES | RED) << 16)), (2, 63)):     # beyond the image's columns, too wide, below the image, a count the mask does not have
        bad = b.copy(); bad[field] = value
        assert lib.hz_blob_scatter(bad.ctypes.data, 64, 8, img.ctypes.data, out.ctypes.data, None, None) == -1, field
