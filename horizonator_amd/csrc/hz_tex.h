/* hz_tex.h - the texture path of the reference ("next" row N4), shared by the
 * HIP kernels and the host.
 *
 * Reference: vertex.glsl:41-61,116-126 (texture coordinates of a vertex),
 * fragment.glsl:17-22 (0.7*texture + 0.3*shade), horizonator-lib.c:247-266
 * (GL_LINEAR, GL_REPEAT, GL_RGB texture of NtilesX*256 x NtilesY*256 texels).
 * As with the rest of the pipeline the arithmetic is what the reference's
 * shaders do ON MESA LLVMPIPE, pinned by probes:
 *   - texture coordinates in the operation order of Mesa's compiler
 *   - sampler: coordinate -> 24.8 fixed point (round to nearest even of
 *     s*size*256, minus half a texel), texel pair weights 0..255, two 8-bit
 *     lerps along x then one along y, each (w*(b-a)+128)>>8; the sampler's
 *     result is byte*(1/255)
 *   - blend in float32 with separate multiplies and add, then round(x*255)
 */
#pragma once

#include "hz_num.h"
#include "hz_hip.h"             /* hz_texparams_t */

/* reference vertex.glsl:116-126 with get_xtexture / get_ytexture (:53-61) */
HZ_HD void hz_vertex_tex(const hz_texparams_t* t, float deg_per_cell, float fi, float fj, float* s, float* tt)
{
    const float lat  = (t->origin_cell_lat_deg + fj*deg_per_cell) * HZ_DEG2RAD;
    const float dlat = lat + -t->viewer_lat_rad;
    const float lon  = (t->origin_cell_lon_deg + fi*deg_per_cell) * HZ_DEG2RAD;
    const float xt   = t->lon1*lon + t->lon0;
    *s  = (xt + -(float)t->lowest_x) / (float)t->ntiles_x;
    const float yt   = dlat*(dlat*t->dlat2 + t->dlat1) + t->dlat0;
    *tt = 1.0f + -((yt + -(float)t->lowest_y) / (float)t->ntiles_y);
}

/* GL_REPEAT + GL_LINEAR along one axis: the two texel indices and the weight
 * (0..255) of the second */
HZ_HD void hz_tex_wrap_linear(float s, int size, int* i0, int* i1, int* w)
{
    const int pot = (size & (size-1)) == 0;
    if(!pot)
    {
        s = s - __builtin_floorf(s);
        if(!(s <= 0.99999994f)) s = 0.99999994f;
    }
    const int fixed = (int)hz_roundeven((s*(float)size)*256.0f) - 128;
    const int ip = fixed >> 8;
    *w = fixed & 255;
    if(pot) { *i0 = ip & (size-1); *i1 = (ip+1) & (size-1); }
    else    { *i0 = ip < 0 ? size-1 : ip; *i1 = ip+1 > size-1 ? 0 : ip+1; }
}

/* texels: one uint32 per texel, B | G<<8 | R<<16, row 0 = texture coordinate
 * t = 0.  Returns the sampled colour in the same packing. */
HZ_HD uint32_t hz_tex_sample(const uint32_t* texels, int tex_w, int tex_h, float s, float tt)
{
    int i0, i1, wx, j0, j1, wy;
    hz_tex_wrap_linear(s,  tex_w, &i0, &i1, &wx);
    hz_tex_wrap_linear(tt, tex_h, &j0, &j1, &wy);
    const uint32_t t00 = texels[(size_t)j0*tex_w + i0], t10 = texels[(size_t)j0*tex_w + i1];
    const uint32_t t01 = texels[(size_t)j1*tex_w + i0], t11 = texels[(size_t)j1*tex_w + i1];
    uint32_t out = 0;
    #pragma unroll
    for(int c=0; c<3; c++)
    {
        const int v00 = (int)((t00 >> (8*c)) & 255u), v10 = (int)((t10 >> (8*c)) & 255u);
        const int v01 = (int)((t01 >> (8*c)) & 255u), v11 = (int)((t11 >> (8*c)) & 255u);
        const int a = v00 + ((wx*(v10 - v00) + 128) >> 8);
        const int b = v01 + ((wx*(v11 - v01) + 128) >> 8);
        out |= (uint32_t)(a + ((wy*(b - a) + 128) >> 8)) << (8*c);
    }
    return out;
}

HZ_HD uint32_t hz_unorm8(float x)
{
    x = hz_max(hz_min(x, 1.0f), 0.0f);
    return (uint32_t)hz_roundeven(x * 255.f);
}

/* reference fragment.glsl:17-22; `texel` as hz_tex_sample returns it, `shade`
 * the interpolated rgb.r; result B | G<<8 | R<<16 */
HZ_HD uint32_t hz_fragment_textured(uint32_t texel, float shade)
{
    const float b = 0.7f*((float)(texel & 255u)         * (1.0f/255.0f));
    const float g = 0.7f*((float)((texel >> 8) & 255u)  * (1.0f/255.0f));
    const float r = 0.7f*((float)((texel >> 16) & 255u) * (1.0f/255.0f)) + 0.3f*shade;
    return hz_unorm8(b) | (hz_unorm8(g) << 8) | (hz_unorm8(r) << 16);
}
