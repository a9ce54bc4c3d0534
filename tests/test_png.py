"""hz_png.c: the PNG reader behind render_texture=true (the reference uses FreeImage, reference
horizonator-lib.c:323-369).  Every flavour a tile server emits must decode to the pixels PIL sees;
what is not supported must fail with a message, not with garbage."""
import ctypes as C

import numpy as np
import pytest

from horizonator_amd import _lib as hzlib

PIL = pytest.importorskip("PIL.Image")


def _load(path, w, h):
    lib = C.CDLL(hzlib.LIB_PATH)
    lib.hz_png_load_rgb.restype = C.c_int
    lib.hz_png_load_rgb.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
    rgb = np.full((h, w, 3), 123, np.uint8)
    err = C.create_string_buffer(512)
    rc = lib.hz_png_load_rgb(str(path).encode(), w, h, rgb.ctypes.data, err, len(err))
    return rc, rgb, err.value.decode()


def _image(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([(xx * 5 + seed) % 256, (yy * 3) % 256, (xx + yy) % 256], -1).astype(int)
    return (a + rng.integers(-30, 31, a.shape)).clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize("flavour", ["rgb", "rgba", "grey", "palette256", "palette16", "palette4", "palette2",
                                     "rgb_no_filter_choice", "rgb_level0"])
def test_supported_flavours_decode_like_pil(tmp_path, flavour):
    w, h = 256, 256
    img = PIL.fromarray(_image(w, h, 5), "RGB")
    kw = {}
    if flavour == "rgba":
        img = img.convert("RGBA")
    elif flavour == "grey":
        img = img.convert("L")
    elif flavour.startswith("palette"):
        img = img.quantize(colors=int(flavour[7:]))
    elif flavour == "rgb_level0":
        kw = dict(compress_level=0)
    elif flavour == "rgb_no_filter_choice":
        kw = dict(optimize=True)
    p = tmp_path / "t.png"
    img.save(p, **kw)
    rc, rgb, err = _load(p, w, h)
    assert rc == 0, err
    assert np.array_equal(rgb, np.asarray(PIL.open(p).convert("RGB")))


def test_odd_width_palette_rows_are_bit_packed(tmp_path):
    w, h = 37, 5                                    # 4-bit palette: 18.5 bytes per row
    img = PIL.fromarray(_image(w, h, 9), "RGB").quantize(colors=16)
    p = tmp_path / "odd.png"
    img.save(p, bits=4)
    rc, rgb, err = _load(p, w, h)
    assert rc == 0, err
    assert np.array_equal(rgb, np.asarray(PIL.open(p).convert("RGB")))


def test_what_is_not_supported_fails_with_a_message(tmp_path):
    w, h = 64, 64
    base = PIL.fromarray(_image(w, h, 1), "RGB")
    # wrong size
    p = tmp_path / "a.png"; base.save(p)
    rc, _, err = _load(p, 256, 256)
    assert rc != 0 and "expected 256x256" in err
    # interlaced (Adam7): a hand-made header is enough, the reader must stop there
    import struct
    import zlib

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)
    pi = tmp_path / "b.png"
    pi.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(bytes(w * h * 3 + h))) + chunk(b"IEND", b""))
    rc, _, err = _load(pi, w, h)
    assert rc != 0 and "interlaced" in err
    # 16 bits per sample
    ihdr16 = struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)
    p16 = tmp_path / "c.png"
    p16.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr16) + chunk(b"IDAT", zlib.compress(bytes(w * h * 2 + h))) + chunk(b"IEND", b""))
    rc, _, err = _load(p16, w, h)
    assert rc != 0 and "not supported" in err
    # no PNG at all, truncated PNG, missing file
    q = tmp_path / "d.png"; q.write_bytes(b"definitely not a png, but long enough to pass the size check " * 3)
    rc, _, err = _load(q, w, h)
    assert rc != 0 and "not a PNG" in err
    good = (tmp_path / "a.png").read_bytes()
    t = tmp_path / "e.png"; t.write_bytes(good[:len(good) // 2])
    rc, _, err = _load(t, w, h)
    assert rc != 0 and err
    rc, _, err = _load(tmp_path / "nothing.png", w, h)
    assert rc != 0 and "cannot open" in err
