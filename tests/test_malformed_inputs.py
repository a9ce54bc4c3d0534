"""Files from disk that are not what they should be: damaged and truncated PNG tiles, .hgt tiles
of the wrong size, directories without tiles.  The readers must answer with an error (or, for a
missing DEM tile, with sea level - reference dem.c:199-206), never with a crash or a read outside
their buffers; tests/test_sanitizers.py runs this file once more under AddressSanitizer."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

import hzutil
from horizonator_amd import _lib as hzlib

PIL = pytest.importorskip("PIL.Image")


def _load_png(path, w, h):
    lib = C.CDLL(hzlib.LIB_PATH)
    lib.hz_png_load_rgb.restype = C.c_int
    lib.hz_png_load_rgb.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
    # the destination is exactly as large as promised, inside a numpy array: a write past it is ASan's to find
    rgb = np.zeros((h, w, 3), np.uint8)
    err = C.create_string_buffer(512)
    rc = lib.hz_png_load_rgb(str(path).encode(), w, h, rgb.ctypes.data, err, len(err))
    return rc, err.value.decode()


def _good_png(tmp_path, mode):
    w = h = 64
    yy, xx = np.mgrid[0:h, 0:w]
    a = (np.stack([(xx * 5) % 256, (yy * 3) % 256, (xx + yy) % 256], -1) + np.random.default_rng(3).integers(0, 40, (h, w, 3))).clip(0, 255).astype(np.uint8)
    img = PIL.fromarray(a, "RGB")
    if mode == "palette":
        img = img.quantize(colors=16)
    elif mode == "rgba":
        img = img.convert("RGBA")
    p = tmp_path / f"good_{mode}.png"
    img.save(p, **({"bits": 4} if mode == "palette" else {}))
    return p.read_bytes(), w, h


@pytest.mark.parametrize("mode", ["rgb", "palette", "rgba"])
def test_truncated_png_at_every_length(tmp_path, mode):
    good, w, h = _good_png(tmp_path, mode)
    p = tmp_path / "t.png"
    for n in list(range(0, 80)) + list(range(80, len(good), 7)):
        p.write_bytes(good[:n])
        rc, err = _load_png(p, w, h)
        if n < len(good) - 12:                      # (a file that only lacks its IEND chunk still decodes)
            assert rc != 0 and err, n


@pytest.mark.parametrize("mode", ["rgb", "palette", "rgba"])
def test_png_with_damaged_bytes(tmp_path, mode):
    """seeded single- and multi-byte damage anywhere in the file (headers, lengths, palette, compressed
    stream, checksums): either an error or - where only pixel data or a CRC was hit - a decode"""
    good, w, h = _good_png(tmp_path, mode)
    rng = np.random.default_rng(7)
    p = tmp_path / "d.png"
    for k in range(400):
        b = bytearray(good)
        for _ in range(1 + k % 3):
            at = int(rng.integers(0, len(b)))
            b[at] = int(rng.integers(0, 256))
        p.write_bytes(bytes(b))
        rc, err = _load_png(p, w, h)
        assert rc == 0 or err


def test_png_chunk_lengths_that_lie(tmp_path):
    def chunk(kind, data, length=None):
        return struct.pack(">I", len(data) if length is None else length) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    w = h = 16
    sig = b"\x89PNG\r\n\x1a\n"
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)
    raw = zlib.compress(bytes((w * 3 + 1) * h))
    cases = {
        "idat_longer_than_file": sig + chunk(b"IHDR", ihdr) + chunk(b"IDAT", raw, length=0x7FFFFFF0) + chunk(b"IEND", b""),
        "idat_length_wraps": sig + chunk(b"IHDR", ihdr) + chunk(b"IDAT", raw, length=0xFFFFFFFF) + chunk(b"IEND", b""),
        "huge_palette": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 3, 0, 0, 0)) + chunk(b"PLTE", bytes(3 * 300))
                        + chunk(b"IDAT", zlib.compress(bytes((w + 1) * h))) + chunk(b"IEND", b""),
        "no_ihdr": sig + chunk(b"IDAT", raw) + chunk(b"IEND", b"") + bytes(40),
        "no_idat": sig + chunk(b"IHDR", ihdr) + chunk(b"IEND", b"") + bytes(40),
        "too_little_data": sig + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(bytes(10))) + chunk(b"IEND", b""),
        "too_much_data": sig + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(bytes((w * 3 + 1) * h * 4))) + chunk(b"IEND", b""),
        "bad_filter_type": sig + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(bytes([9]) + bytes((w * 3 + 1) * h - 1))) + chunk(b"IEND", b""),
        "palette_index_out_of_range": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 3, 0, 0, 0)) + chunk(b"PLTE", bytes(6))
                                      + chunk(b"IDAT", zlib.compress((b"\x00" + b"\xff" * w) * h)) + chunk(b"IEND", b""),
    }
    for name, data in cases.items():
        p = tmp_path / (name + ".png")
        p.write_bytes(data)
        rc, err = _load_png(p, w, h)
        if name in ("huge_palette", "palette_index_out_of_range"):
            assert rc == 0, (name, err)             # decodable: entries beyond the palette read as black
        else:
            assert rc != 0 and err, name


def _dem_init(d, lat=34.5, lon=-117.5, R=100, srtm1=False):
    lib = hzlib.load()
    ctx = hzlib.DemContext()
    ok = lib.horizonator_dem_init(C.byref(ctx), lat, lon, R, -1.0, str(d).encode(), srtm1)
    z = None
    if ok:
        z = [lib.horizonator_dem_sample(C.byref(ctx), i, j) for i in (0, 1, 57, 2 * R - 1) for j in (0, 33, 2 * R - 1)]
        lib.horizonator_dem_deinit(C.byref(ctx))
    return bool(ok), z


def test_hgt_tiles_of_the_wrong_size(tmp_path):
    """reference dem.c:234-239: a tile whose size is not (cpd+1)^2*2 bytes makes init fail;
    dem.c:199-206: a missing tile is sea level"""
    ok, z = _dem_init(tmp_path)                     # no tiles at all
    assert ok and set(z) == {0}
    name = "N34W118.hgt"
    full = 1201 * 1201 * 2
    for size in (0, 1, 2, full - 2, full - 1, full + 1, full + 2, 3601 * 3601 * 2):
        (tmp_path / name).write_bytes(bytes(size))
        ok, z = _dem_init(tmp_path)
        if size == 0:
            assert ok and set(z) == {0}             # an empty file counts as missing (reference dem.c:213-221)
        else:
            assert not ok, size
    (tmp_path / name).write_bytes(b"\x00\x07" * (1201 * 1201))
    ok, z = _dem_init(tmp_path)
    assert ok and set(z) == {7}
    # an SRTM3-sized file offered as SRTM1
    ok, _ = _dem_init(tmp_path, srtm1=True)
    assert not ok


def test_hgt_directory_that_is_not_one(tmp_path):
    f = tmp_path / "file"
    f.write_bytes(b"x")
    ok, z = _dem_init(f)                            # tiles "inside" a regular file: all missing -> sea level
    assert ok and set(z) == {0}
    ok, z = _dem_init(tmp_path / "does" / "not" / "exist")
    assert ok and set(z) == {0}
