"""TEST INFRASTRUCTURE: drive oracle/_ref/glsl_golden (the reference's shaders on
Mesa llvmpipe).  Works only where /root/reference and Mesa's swrast driver exist
(the build container); nothing at test time on the GPU box calls this."""
import os
import struct
import subprocess
import tempfile

import numpy as np

from . import GLSL_GOLDEN_PATH, VIEW_FIELDS

REFERENCE_DIR = os.environ.get("HZ_REFERENCE_DIR", "/root/reference")

# order of the uniform block in the job file (glsl_golden.c header)
_JOB_UNIFORMS = ("viewer_cell_i", "viewer_cell_j", "viewer_z", "deg_per_cell", "cos_viewer_lat",
                 "az_deg0", "az_deg1", "aspect", "znear", "zfar", "znear_color", "zfar_color")


def available():
    return os.path.exists(GLSL_GOLDEN_PATH) and os.path.exists(os.path.join(REFERENCE_DIR, "vertex.glsl"))


def _run(mode, N, W, H, view, body, threads=None):
    u = np.array([getattr(view, n) for n in _JOB_UNIFORMS], np.float32)
    with tempfile.TemporaryDirectory() as td:
        job, res = os.path.join(td, "job"), os.path.join(td, "res")
        with open(job, "wb") as f:
            f.write(struct.pack("<4i", mode, N, W, H))
            f.write(u.tobytes())
            f.write(body)
        env = dict(os.environ)
        if threads is not None:
            env["LP_NUM_THREADS"] = str(threads)
        p = subprocess.run([GLSL_GOLDEN_PATH, REFERENCE_DIR, job, res], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if p.returncode != 0:
            raise RuntimeError("glsl_golden failed: " + p.stderr.decode())
        return np.fromfile(res, np.uint8), p.stderr.decode()


def _split_frame(raw, W, H):
    n = W * H
    bgr = raw[:3 * n].reshape(H, W, 3)
    depth = raw[3 * n:7 * n].view(np.float32).reshape(H, W)
    z24 = raw[7 * n:11 * n].view(np.uint32).reshape(H, W)
    # GL rows are bottom-first; flip to the top-first convention of the outputs
    return {"bgr": bgr[::-1].copy(), "depth": depth[::-1].copy(), "z24": z24[::-1].copy()}


def render(mosaic, view, W, H, threads=None):
    """the reference's draw: dict(bgr uint8[H,W,3], depth float32[H,W], z24 uint32[H,W]), top row first"""
    mosaic = np.ascontiguousarray(mosaic, np.int16)
    raw, log = _run(0, mosaic.shape[0], W, H, view, mosaic.tobytes(), threads)
    out = _split_frame(raw, W, H)
    out["log"] = log
    return out


def vertices(mosaic, view):
    """the reference's vertex shader on every vertex: float32[N,N,5] = gl_Position.xyzw, rgb.r"""
    mosaic = np.ascontiguousarray(mosaic, np.int16)
    N = mosaic.shape[0]
    raw, _ = _run(1, N, 4, 4, view, mosaic.tobytes())
    return raw.view(np.float32).reshape(N, N, 5)


def raw_triangles(tris, W, H):
    """probe of llvmpipe's rasteriser with pass-through shaders of our own:
    tris float32[n,3,4] = clip x,y,z + red per vertex"""
    tris = np.ascontiguousarray(tris, np.float32)

    class _V:
        pass
    v = _V()
    for n in VIEW_FIELDS:
        setattr(v, n, 0.0)
    raw, log = _run(2, 0, W, H, v, struct.pack("<i", tris.shape[0]) + tris.tobytes())
    return _split_frame(raw, W, H)


# ---- the texture path (reference vertex.glsl:41-61,116-126, fragment.glsl:17-22) ----

TEX_FLOATS = ("viewer_lat_rad", "origin_cell_lon_deg", "origin_cell_lat_deg",
              "texturemap_lon0", "texturemap_lon1", "texturemap_dlat0", "texturemap_dlat1", "texturemap_dlat2")
TEX_INTS = ("NtilesX", "NtilesY", "osmtile_lowestX", "osmtile_lowestY")


def _tex_block(tex, texels):
    """tex: dict with TEX_FLOATS and TEX_INTS; texels uint8[texH,texW,3] exactly as the
    reference hands them to glTexSubImage2D(GL_BGR): row 0 first (GL: bottom), B,G,R"""
    texels = np.ascontiguousarray(texels, np.uint8)
    th, tw, _ = texels.shape
    return (np.array([tex[k] for k in TEX_FLOATS], np.float32).tobytes()
            + struct.pack("<6i", *[int(tex[k]) for k in TEX_INTS], tw, th) + texels.tobytes())


def render_textured(mosaic, view, W, H, tex, texels, threads=None):
    mosaic = np.ascontiguousarray(mosaic, np.int16)
    raw, log = _run(3, mosaic.shape[0], W, H, view, _tex_block(tex, texels) + mosaic.tobytes(), threads)
    out = _split_frame(raw, W, H)
    out["log"] = log
    return out


def vertices_textured(mosaic, view, tex, texels=None):
    """float32[N,N,7] = gl_Position.xyzw, rgb.r, tex.xy"""
    mosaic = np.ascontiguousarray(mosaic, np.int16)
    N = mosaic.shape[0]
    if texels is None:
        texels = np.zeros((2, 2, 3), np.uint8)
    raw, _ = _run(4, N, 4, 4, view, _tex_block(tex, texels) + mosaic.tobytes())
    return raw.view(np.float32).reshape(N, N, 7)


def textured_triangles(tris, W, H, texels, NtilesX=1, NtilesY=1, sampler_only=False):
    """probe of llvmpipe's texture sampling through the reference's fragment shader:
    tris float32[n,3,6] = clip x,y,z, red, s, t per vertex"""
    tris = np.ascontiguousarray(tris, np.float32)

    class _V:
        pass
    v = _V()
    for n in VIEW_FIELDS:
        setattr(v, n, 0.0)
    tex = {k: 0.0 for k in TEX_FLOATS}
    tex.update(NtilesX=NtilesX, NtilesY=NtilesY, osmtile_lowestX=0, osmtile_lowestY=0)
    raw, log = _run(6 if sampler_only else 5, 0, W, H, v,
                    _tex_block(tex, texels) + struct.pack("<i", tris.shape[0]) + tris.tobytes())
    return _split_frame(raw, W, H)
