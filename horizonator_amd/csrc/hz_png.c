/* hz_png.c - see hz_png.h */
#include "hz_png.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define FAIL(...) do { snprintf(err, errlen, __VA_ARGS__); goto done; } while(0)

static uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

int hz_png_load_rgb(const char* path, int width, int height, unsigned char* rgb, char* err, size_t errlen)
{
    int result = -1;
    unsigned char *file = NULL, *idat = NULL, *raw = NULL;
    unsigned char palette[256][3];
    int npal = 0;
    memset(palette, 0, sizeof(palette));

    FILE* f = fopen(path, "rb");
    if(!f) { snprintf(err, errlen, "cannot open '%s'", path); return -1; }
    fseek(f, 0, SEEK_END);
    const long nfile = ftell(f);
    fseek(f, 0, SEEK_SET);
    if(nfile < 8+25+12) { fclose(f); snprintf(err, errlen, "'%s' is not a PNG file", path); return -1; }
    file = malloc((size_t)nfile);
    if(!file || fread(file, 1, (size_t)nfile, f) != (size_t)nfile) { fclose(f); free(file); snprintf(err, errlen, "cannot read '%s'", path); return -1; }
    fclose(f);

    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if(memcmp(file, sig, 8) != 0) FAIL("'%s' is not a PNG file", path);

    int depth = 0, ctype = -1;
    size_t nidat = 0;
    idat = malloc((size_t)nfile);
    if(!idat) FAIL("out of memory");
    for(size_t at = 8; at + 12 <= (size_t)nfile; )
    {
        const uint32_t len = be32(file + at);
        const unsigned char* type = file + at + 4;
        const unsigned char* data = file + at + 8;
        if(at + 12 + (size_t)len > (size_t)nfile) FAIL("'%s': truncated chunk", path);
        if(!memcmp(type, "IHDR", 4))
        {
            if(len != 13) FAIL("'%s': bad IHDR", path);
            if((int)be32(data) != width || (int)be32(data+4) != height)
                FAIL("'%s' is %ux%u, expected %dx%d", path, be32(data), be32(data+4), width, height);
            depth = data[8]; ctype = data[9];
            if(data[10] != 0 || data[11] != 0) FAIL("'%s': unknown compression/filter method", path);
            if(data[12] != 0) FAIL("'%s': interlaced PNGs are not supported", path);
        }
        else if(!memcmp(type, "PLTE", 4))
        {
            npal = (int)(len/3);
            if(npal > 256) npal = 256;
            memcpy(palette, data, (size_t)npal*3);
        }
        else if(!memcmp(type, "IDAT", 4)) { memcpy(idat + nidat, data, len); nidat += len; }
        else if(!memcmp(type, "IEND", 4)) break;
        at += 12 + (size_t)len;
    }
    int channels;
    switch(ctype)
    {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 6: channels = 4; break;
    default: FAIL("'%s': colour type %d is not supported", path, ctype);
    }
    if(!((ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) || (ctype != 3 && depth == 8)))
        FAIL("'%s': %d bits per sample with colour type %d is not supported", path, depth, ctype);

    const size_t stride = ((size_t)width*channels*depth + 7)/8;     /* bytes per scanline, without the filter byte */
    const int    bpp    = (channels*depth + 7)/8;                   /* filter distance */
    uLongf nraw = (uLongf)((stride + 1)*(size_t)height);
    raw = malloc(nraw);
    if(!raw) FAIL("out of memory");
    if(uncompress(raw, &nraw, idat, (uLong)nidat) != Z_OK || nraw != (stride + 1)*(size_t)height)
        FAIL("'%s': corrupt image data", path);

    /* undo the scanline filters in place */
    for(int y=0; y<height; y++)
    {
        unsigned char* cur = raw + (size_t)y*(stride+1) + 1;
        const unsigned char* up = y ? cur - (stride+1) : NULL;
        const int ft = cur[-1];
        for(size_t x=0; x<stride; x++)
        {
            const int a = x >= (size_t)bpp ? cur[x-bpp] : 0;
            const int b = up ? up[x] : 0;
            const int c = (up && x >= (size_t)bpp) ? up[x-bpp] : 0;
            int v = cur[x];
            switch(ft)
            {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b)/2; break;
            case 4: v += paeth(a, b, c); break;
            default: FAIL("'%s': unknown filter type %d", path, ft);
            }
            cur[x] = (unsigned char)v;
        }
    }

    for(int y=0; y<height; y++)
    {
        const unsigned char* cur = raw + (size_t)y*(stride+1) + 1;
        unsigned char* out = rgb + (size_t)y*width*3;
        for(int x=0; x<width; x++)
        {
            if(ctype == 3)
            {
                const int bit = x*depth;
                const int idx = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                out[3*x+0] = palette[idx][0]; out[3*x+1] = palette[idx][1]; out[3*x+2] = palette[idx][2];
            }
            else if(ctype == 0) { out[3*x+0] = out[3*x+1] = out[3*x+2] = cur[x]; }
            else { out[3*x+0] = cur[x*channels]; out[3*x+1] = cur[x*channels+1]; out[3*x+2] = cur[x*channels+2]; }
        }
    }
    result = 0;

 done:
    free(file); free(idat); free(raw);
    return result;
}
