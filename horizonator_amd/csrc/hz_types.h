/* hz_types.h - what the kernels (hz_kernels.hip and its hz_k_*.h) and the host code that launches them (hz_draw.cpp, hz_convert.cpp,
 * hz_hostpath.cpp) share: kernel parameters, the records and queues between the kernels of a draw, the constants both
 * sides derive launch grids and buffer sizes from.  Plain data and host+device helpers only; no device code. */
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "hz_raster.h"          /* hz_xform_t (hz_num.h), hz_edges_t */
#include "hz_tex.h"
#include "hz_scatter.h"

/* ------------------------------------------------------------------------ */
/* kernel parameters                                                         */

/* coarse depth (hz_k_hiz.h): the largest upper half of a framebuffer word per 8 x 4 (l1) and 32 x 16 (l2) pixels,
 * w1 / w2 tiles per row; l1 == NULL: the draw has none */
struct hz_hiz_t { uint32_t* l1; uint32_t* l2; int w1, w2; };

/* whose fragments an experiment of the -DHZ_EXPERIMENTS build acts on (hz_params_t::exp_fb) */
#define HZ_WHO_MARCH 0
#define HZ_WHO_BIG   1
#define HZ_WHO_OTHER 2                  /* (k_clip's own fragments: never part of an experiment) */

struct hz_params_t
{
    hz_xform_t u;
    float halfW, halfH;
    int   N;                /* samples per mosaic axis                  */
    int   W, H;             /* full image size                          */
    int   col0, col1;       /* sector [col0,col1)                       */
    int   SW;               /* col1-col0, row stride of fb              */
    unsigned long long* wave_cycles;   /* diagnostics: per-wave duration of k_march, or NULL */
    unsigned int inline_max;           /* k_march: boxes up to this many pixels are rasterised by the marching wave */
    unsigned int big_min;              /* k_march: boxes above this many pixels go to k_big (tiles), between: k_mid  */
    float far_dd;                      /* k_march: squared horizontal distance beyond which a vertex is surely past zfar */
    int   far_strips;                  /* some vertex of the mosaic lies beyond that: whole strips may (k_march asks) */
    /* two-pass draw (see hz_hip_draw): which strips a k_march launch takes, and
     * whether it tests its survivors against the depth already in the framebuffer */
    const uint32_t* worklist;          /* k_march: the (segment, strip column) pairs of this launch, one per workgroup; NULL = the launch grid says it */
    int   cull_strips;                 /* k_march: strips whose four corners lie outside the drawn columns leave at once (sectors, views < 360 degrees) */
    int   pass;                        /* 0 every strip, 1 only the strips next to the viewer, 2 all the others */
    int   near_x0, near_x1;            /* strip columns [x0,x1] and                                             */
    int   near_j0, near_j1;            /* cell rows [j0,j1) that make up "next to the viewer"                   */
    int   early_z;                     /* mr_flush: skip triangles whose box is already covered by nearer depth */
    int   pretest_march;               /* the marching waves read a word before the atomic (draw_impl decides) */
    int   qshards_log2;                /* the counters the queue of big triangles is appended through: 1 << that (HZ_QSHARDS, below) */
#ifdef HZ_EXPERIMENTS
    int   exp_fb[2];                   /* experiments (wrong pictures), see hz_fb_min: [0] the marching waves' fragments, [1] k_big's */
#endif
    const hz_polar_t* vcache;          /* k_march<.., VCACHE>: the view-independent half of every vertex's transform, [N][N] (hz_num.h: hz_polar_t; hz_draw.cpp: the vertex cache) */
    const uint32_t* hiz;               /* mr_flush, k_big: coarse depth (hz_k_hiz.h: level 1, level 2 behind it; second rounds of zoomed views), or NULL.
                                        * (One pointer, the rest follows from SW and H: every scalar register k_march holds costs it lane spills in its loop.) */
    float z_guard;                     /* hz_tri_depth_floor(): 1/500 + max(W,H)*2^-22                          */
    float z_hide_k;                    /* hz_tri_hidden(): 1.03 * z_guard * (2^24-1)                            */
    int   fast_ok;                     /* hzf_draw_ok(): the uniforms allow the abridged division/sqrt sequences */
    int   quad_max_dx;                 /* k_march: 256*(W/16 - 1): see the cull of whole cells                  */
#ifdef HZ_EXPERIMENTS
    int   debug;                       /* HZ_MARCH_DEBUG (timing splits, wrong pictures): 1 survivors are dropped,
                                        * 2 survivors are dropped after the early depth test */
#endif
    /* one byte per HZ_SEG consecutive pixels of a framebuffer row (row stride
     * seg_stride): nonzero once anything was drawn there.  Every write to the
     * framebuffer sets it (hz_fb_min); the conversion skips reading - and
     * clearing - segments nothing touched: the sky, 62 % of the benchmark's
     * pixels.  A stale nonzero byte only costs the read. */
    unsigned char* touched;
    int   seg_stride;
};
#define HZ_SEG_LOG2 8
#define HZ_SEG      (1 << HZ_SEG_LOG2)

/* a set-up triangle as it travels between phases: through LDS inside
 * k_scatter (stride 23 dwords = odd, conflict-free), through HBM to k_mid and
 * k_big.  Coverage as hz_edges_t: what the pixel loops need, ready made. */
struct hz_rec_t
{
    hz_edges_t e;
    float    z_org, dzdx, dzdy, r_org, drdx, drdy;
    int32_t  px0, py0, bw;
    float    inv_bw;
    uint32_t prim;
};
struct hz_bigrec_t { hz_rec_t r; int32_t bh; };

/* work item of the large-triangle pass: 64 tiles of one triangle */
struct hz_bigitem_t { uint32_t rec; uint32_t chunk; };

/* the HBM queues between the kernels of one draw */
struct mr_queue_t
{
    hz_bigrec_t*  bigrec;           /* set-up triangles for k_big                                */
    hz_bigitem_t* bigitem;          /* ... and their work items                                  */
    hz_rec_t*     midrec;           /* set-up triangles for k_mid                                */
    uint32_t*     clip;             /* ids of triangles that have to go through the clipper      */
    unsigned int* counters;         /* [4] clip ids; the big and the medium triangles' counters:
                                     * HZ_QSHARDS of them from [HZ_QSHARD0] on (below); [0] [1] [2]: their sums as the last draw left them */
    unsigned int  bigrec_capacity, bigitem_capacity, midrec_capacity, clip_capacity;
};

/* The queue of big triangles has HZ_QSHARDS counters (round 5).  Every append is one RETURNING atomic - the appender needs
 * the index it got -, and one address takes 83 M of those a second from the whole chip, 12 ns each, whoever asks
 * (tools/atomic_one_address.hip; 16 addresses: 16 times that): the 10 degree view's second round has 36.6 K flushes that
 * append - 0.44 ms of atomics on ONE word for a kernel of 0.42 ms -, a first round's marching kernel 3.5 K in its 44 us.
 * A wave appends through the counter of its shard (its block's number); the shards own the arrays of records and of items
 * in turns, HZ_QBLOCK slots at a time (16 items are one cache line: handed out one slot at a time the items cost a line
 * each to write and to read, and a render of a series 3 % - profiles/r5_ab_queue_shards.txt): the l-th record or item of
 * shard s lies in slot HZ_QSLOT(l, s, sl), so the arrays stay dense up to S times the longest shard - the consumers walk that
 * far and skip the slots beyond a shard's own count. */
#define HZ_QSHARDS_LOG2  4
#define HZ_QSHARDS       (1 << HZ_QSHARDS_LOG2)
#define HZ_QBLOCK_LOG2   4
#define HZ_QBLOCK        (1 << HZ_QBLOCK_LOG2)
/* (sl: log2 of the shards a draw uses - hz_params_t::qshards_log2: HZ_QSHARDS_LOG2 for zoomed views, whose waves append with nearly
 * every flush; 0 = one counter and the arrays in plain order for whole panoramas, whose appends are few: with sixteen shards a
 * render of a series took 1-3 % longer, profiles/r5_ab_queue_shards.txt) */
#define HZ_QSHARD_ROOM(capacity, sl) ((((capacity) >> ((sl) + HZ_QBLOCK_LOG2))) << HZ_QBLOCK_LOG2)      /* records or items a shard may hold of an array of `capacity` */
#define HZ_QSLOT(l, s, sl) (((((((uint32_t)(l)) >> HZ_QBLOCK_LOG2) << (sl)) + (uint32_t)(s)) << HZ_QBLOCK_LOG2) | (((uint32_t)(l)) & (HZ_QBLOCK-1)))
#define HZ_QSHARD_STRIDE 32             /* words between two shards' counters: 128 bytes */
#define HZ_QSHARD0       16             /* shard s: counters[HZ_QSHARD0 + s*HZ_QSHARD_STRIDE + {0 records, 1 items (one 64-bit word), 2 ~(first invalid item); 4 records for k_mid, 5 ~(first invalid one)}] */
#define HZ_NCOUNTERS (HZ_QSHARD0 + HZ_QSHARDS*HZ_QSHARD_STRIDE)
#define HZ_CNT_LAST  8                  /* [8..14): the counters as the last draw left them (diagnostics; the big triangles': sums over the shards) */
#ifndef HZ_NFB
#define HZ_NFB 3                        /* framebuffers (and queue sets per round) a context cycles through */
#endif
#define HZ_STAGE_SLOTS 32               /* pinned staging chunks in flight between device and caller memory (128 MB: a copy of four chunks must never wait for the scatter of an old one) */
#define HZ_STAGE_BYTES ((size_t)4 << 20)    /* (16 MB until round 5: a chunk's blobs are scattered when all of it has arrived - 0.26 ms for 16 MB) */
#define HZ_COPY_STREAMS 2               /* device -> host copies alternate between that many streams (copy engines) */
#define HZ_HOST_BANDS  4                /* the conversion runs in that many bands of rows when its results go to the host */
#define HZ_INLINE_MAX_PIX  64       /* k_scatter: boxes up to this many pixel centres are rasterised in the block */

/* ------------------------------------------------------------------------ */
/* the marching kernel's strips, segments and work lists (hz_k_march.h)      */

#define MR_COLS   63
/* round 1 of a draw takes the strips within ppr/20 cells of the viewer (draw_impl: plan_rounds); timed at 16000x4000,
 * round 2: 32 cells 1.085 ms per render, 64: 1.063, 96: 1.056, 128: 1.052, 192: 1.064, 256: 1.12 */
/* The cap of the first round's reach (zoomed views and panoramas wider than 49000 columns hit it), and what the first
 * rounds' queue sets are sized for (or HZ_NEAR_CELLS, if larger).  256 until zoomed views got coarse depth (hz_k_hiz.h):
 * a narrow view's frustum is narrow in elevation too, little of the nearest terrain is inside it, and the ridge
 * that hides most of the view tends to lie further out - seven 10 and 45 degree views (three viewpoints, four
 * directions, the rough DEM; tools/hiz_ab.py, profiles/r3_coarse_depth.txt) take 14.5 ms in sum with a reach of 256
 * cells, 11.0 with 384, 11.1 with 512 (five gain up to 2x, two lose 8 %).  Round 4: with the first round's large
 * triangles drawn by screen tile (hz_k_tile.h) a longer first round costs less - the same views 9.7 ms with 384, 9.8 with
 * 448, 9.5 with 512, 9.8 with 576, 10.3 with 640, and the slowest of them 2.03 / 1.95 / 1.72 / 1.69 / 1.80
 * (profiles/r4_tile_batches.txt) - but it is two of the seven that gain (summit, valley: 0.3-0.4 ms each) and five
 * that lose 0.1.  So views that are still "zoomed" at 512 cells (a cell there HZ_HIZ_MIN_PX pixels wide: up to 70
 * degrees at 16000 columns) MAY reach that far: they do when the draws of the same view before them say it pays
 * (hz_draw.cpp, adapt: what the second round had to queue), the others and every first draw of a view 384. */
#define HZ_NEAR_CELLS_WIDE 384
#define HZ_NEAR_CELLS_MAX  512

/* The strips are cut into segments of rows; the segment length depends on the
 * distance (in rows) from the viewer's row so that every wave gets a comparable
 * amount of pixel work: zones south->north with 64, 16, 4, 2, 4, 16, 64 rows
 * per segment.  Built on the host per draw (mr_make_zones).  Measured: with
 * uniform 64-row segments the waves next to the viewer run 10-50x longer than
 * the median and set the kernel time. */
#define MR_NZONES 7
struct mr_zones_t
{
    int row0[MR_NZONES+1];          /* first cell row of each zone; row0[MR_NZONES] = N-1 */
    int rows[MR_NZONES];            /* cell rows per segment                              */
    int seg0[MR_NZONES];            /* number of the zone's first segment                 */
    int nseg[MR_NZONES];            /* segments in the zone                               */
    int total;                      /* all segments = gridDim.y                           */
    int near_first;                 /* dispatch order: segments nearest to the viewer's row first */
};

/* cell rows [jbeg, jend) of segment `seg` (= blockIdx.y of a grid launch, or the
 * segment field of a work-list item); vertex rows jbeg..jend.  Device and host
 * (the work lists of draw_impl) use the same function. */
HZ_HD void mr_segment_rows(const mr_zones_t& zn, int seg, int* jbeg, int* jend)
{
    int zone = 0;
    #pragma unroll
    for(int z=1; z<MR_NZONES; z++)
        if(seg >= zn.seg0[z] && seg < zn.seg0[z] + zn.nseg[z]) zone = z;
    int sseg = seg - zn.seg0[zone];
    if(zn.near_first && zone < MR_NZONES/2) sseg = zn.nseg[zone]-1 - sseg;      /* south of the viewer: northernmost first */
    const int jb = zn.row0[zone] + sseg*zn.rows[zone];
    const int je = jb + zn.rows[zone];
    *jbeg = jb;
    *jend = je < zn.row0[zone+1] ? je : zn.row0[zone+1];
}

/* a work-list item: one marching wave = (segment, strip column) */
#define MR_ITEM_SX_BITS 12
#define MR_ITEM(seg, sx) (((uint32_t)(seg) << MR_ITEM_SX_BITS) | (uint32_t)(sx))

/* ------------------------------------------------------------------------ */
/* the tile-binned rasteriser (hz_k_tile.h), coarse depth (hz_k_hiz.h), k_scatter's blocks, conversions */

#define TL_W 64
#define TL_H 64

#define TL_LIST 2048                /* triangles a tile's list holds */
#define TL_BATCH 64                 /* triangles a workgroup draws into its LDS tile before it merges the tile into the framebuffer */
#define TL_ROW_MIN 24               /* average pixels per non-empty row, within the tile, from which a triangle is drawn row by row (16..48: the same within 5 %) */
#define TL_UNITS_PER_TILE 4         /* room in the unit list, per tile of the image (busy tiles are a fraction, most of them with one batch) */

/* what the tile kernels share: per queue set, allocated with the context */
struct tl_bins_t
{
    unsigned int* cursor;           /* [ntiles]: triangles listed for the tile (zeroed in front of k_tile_bin)                        */
    unsigned int* pairs;            /* [ntiles][TL_LIST]: record numbers                                                          */
    unsigned int* state;            /* [0] 1 = a list (or the unit list) overflowed: k_tile_raster stands down, k_big draws the round;
                                     * [1] units of work (both zeroed in front of k_tile_bin)                                     */
    unsigned int* busy;             /* [TL_UNITS_PER_TILE*ntiles]: the units of work, tile | batch << 24                          */
    unsigned int  units_cap;
    int           tiles_x, tiles_y;
    unsigned int  list_cap;         /* <= TL_LIST (tests make it small: HZ_TILE_LIST) */
};

#define HIZ1_W_LOG2 3
#define HIZ1_H_LOG2 2
#define HIZ2_W_LOG2 5
#define HIZ2_H_LOG2 4
#define HIZ_UNIT_ROWS 16                /* one wave sweeps 256 columns (a segment of hz_params_t::touched) x 16 rows */

/* tiles per row / rows of tiles of the two levels of a framebuffer of SW x H */
HZ_HD size_t hiz_w1(int SW) { return (size_t)((SW + (1 << HIZ1_W_LOG2) - 1) >> HIZ1_W_LOG2); }
HZ_HD size_t hiz_w2(int SW) { return (size_t)((SW + (1 << HIZ2_W_LOG2) - 1) >> HIZ2_W_LOG2); }
HZ_HD size_t hiz_h1(int H)  { return (size_t)((H  + (1 << HIZ1_H_LOG2) - 1) >> HIZ1_H_LOG2); }
HZ_HD size_t hiz_h2(int H)  { return (size_t)((H  + (1 << HIZ2_H_LOG2) - 1) >> HIZ2_H_LOG2); }
/* words of both levels for an image of W x H (level 2 behind level 1) */
HZ_HD size_t hiz_words(int W, int H) { return hiz_w1(W)*hiz_h1(H) + hiz_w2(W)*hiz_h2(H); }

#define SC_CX 64
#define SC_CY 4
#define SC_VX (SC_CX+1)
#define SC_VY (SC_CY+1)
#define SC_THREADS (SC_CX*SC_CY)

#define TX_SUB   4                      /* sub-spans of 64 pixels per chunk */
#define TX_CHUNK (64*TX_SUB)

/* results for host memory: where k_pack_host writes its stream of blobs (hz_k_resolve.h) */
struct hz_hostpack_t
{
    uint32_t*     out;              /* the stream                                                              */
    unsigned int* cursor;           /* [0] words of the stream in use, [1] blobs, [2] nonzero: a blob did not fit */
    unsigned int  capacity;         /* words                                                                   */
    unsigned int  chunk_words;      /* no blob straddles a multiple of this                                    */
    uint32_t      flags;            /* HZ_BLOB_*: the arrays a blob carries                                    */
    unsigned int* present;          /* one bit per tile of the grid (tile = blockIdx.y*gridDim.x + blockIdx.x): a blob was sent for it; may be NULL */
};
#define HP_NONE 0xFFFFFFFFu

/* ... and what k_tell (hz_k_tell.h) tells the host about that stream, in pinned host memory */
struct hz_tell_t
{
    const unsigned int* cursor;         /* k_pack_host's cursor words: [0] words in use, [1] blobs, [2] overflow      */
    const unsigned int* present;        /* the sector's tile bitmap in HBM, npresent words                            */
    unsigned int*       h_info;         /* pinned: 4 words                                                            */
    unsigned int*       h_present;      /* pinned: npresent words                                                     */
    unsigned int        capacity;       /* words of the stream                                                        */
    unsigned int        npresent;
    unsigned int        epoch;          /* never 0                                                                    */
};


#define SP_WAVES  4                     /* rows per workgroup                            */
#define SP_STEPS  8                     /* steps whose words stay in registers           */
#define SP_MAXIT  256                   /* steps per row: sectors up to 65536 columns    */

/* the gathered strips of a panorama for k_resolve_sparse */
#define HZ_MAX_STRIPS 16
struct hz_strips_t
{
    const uint32_t* in[HZ_MAX_STRIPS];
    int ncols[HZ_MAX_STRIPS], col0[HZ_MAX_STRIPS];
};
