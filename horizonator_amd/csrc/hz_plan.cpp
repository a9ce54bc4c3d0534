/* hz_plan.cpp - the plan of a draw, host arithmetic only (no HIP call in this file): the kernel parameters of a view
 * (hz_make_params), the zones of rows a marching wave walks (hz_make_zones), one round or two and the first round's reach
 * (hz_plan_rounds), and the work lists of sectors and views of less than the full circle - which strips of the DEM can
 * reach the drawn columns (hz_list_items).  Used by hz_draw.cpp, hz_hostpath.cpp and the diagnostics of the self-test
 * library; tests/test_worklist.py checks the lists on the CPU. */
#include "hz_dev.h"
#include "hz_fast.h"

/* segment zones of k_march for this view: a cell `r` rows away from the viewer
 * is about ppr/r pixels wide (ppr = pixels per radian of azimuth) */
mr_zones_t hz_make_zones(const hz_params_t& p, bool near_first)
{
    const float ppr = p.halfW * p.u.az_ndc_per_rad;
    const int   ncr = p.N-1;                                /* cell rows */
    const float vj  = p.u.viewer_cell_j;
    const int r2  = (int)(ppr/16.f) + 1;                    /* cells wider than ~16 px: 2-row segments */
    const int r4  = (int)(ppr/4.f) + 1;                     /* ~4 px: 4-row segments                   */
    const int r16 = (int)(ppr/1.f) + 1;                     /* ~1 px: 16-row segments                  */
    auto clampi = [&](float x) { int v = (int)floorf(x); if(v < 0) v = 0; if(v > ncr) v = ncr; return v; };
    mr_zones_t z;
    z.row0[0] = 0;
    z.row0[1] = clampi(vj - (float)r16);
    z.row0[2] = clampi(vj - (float)r4);
    z.row0[3] = clampi(vj - (float)r2);
    z.row0[4] = clampi(vj + (float)r2 + 1.f);
    z.row0[5] = clampi(vj + (float)r4 + 1.f);
    z.row0[6] = clampi(vj + (float)r16 + 1.f);
    z.row0[7] = ncr;
    /* a narrow azimuth sector keeps only a fraction of the waves alive: shorter
     * segments far from the viewer then restore the parallelism (at the price
     * of one extra vertex row per segment) */
    /* (32 rows at most since round 5: the far zones are dispatched last in draws with the early depth test, and with 64 rows
     * their waves - 100 us each - were the kernel's tail: whole panorama, k_march alone 0.623 -> 0.613 ms, a render of a
     * series 0.844 -> 0.837, two alternating runs; 16 rows: 0.616 / 0.842) */
    int far_rows = 64*p.SW/p.W;
    if(far_rows < 16) far_rows = 16;
    if(far_rows > 32) far_rows = 32;
    /* ... and nearer in (cells of 1 to 4 pixels) a narrow sector's kernel was as long as its longest waves: 16 rows of
     * 63 cells with a visible triangle in nearly every lane and a flush per row take 100-170 us (tools/wave_timing.py,
     * HZ_WT_SECTOR=8,0), the whole sector's waves 92 us of the chip - the kernel took 174.  Sectors of less than a sixth
     * of the image cut that zone into 8-row segments: an eighth's strips back to back 0.198 -> 0.169 ms (the widest),
     * 0.156 -> 0.153 (the narrowest); a quarter's and the whole image's waves are many enough to hide their longest
     * (0.273 -> 0.275; profiles/r4_sector_rules.txt). */
    const int z16 = 6*p.SW < p.W ? 8 : 16;          /* (a whole panorama with 8: the kernel +3 %; with 32: -1 % alone, the same in a series - profiles/r5_ab_march_loop.txt (8)) */
    /* A far clip so close that even the farthest cell is four pixels wide (the API's default 40 km at 16000 columns) leaves a
     * second round fewer waves than the chip has slots for, and the kernel is as long as the longest of them (tools/wave_timing.py,
     * HZ_WT_ZFAR=40000: 3.8 K waves, 106 us of work per slot, the longest wave 379 us): two rows to a wave there instead of four -
     * the kernel 0.141 -> 0.105 ms, a render that is waited for 0.896 -> 0.861, a render of a series 0.589 -> 0.583 (its first
     * round is what a series at 40 km waits for); the whole panorama at 600 km with 2: 0.829 -> 0.850. */
    const float cells_to_zfar = sqrtf(p.far_dd)/(p.u.deg_per_cell*111194.9f);
    const int z4 = cells_to_zfar <= 0.25f*ppr ? 2 : 4;
    int rows[MR_NZONES] = { far_rows, z16, z4, 2, z4, z16, far_rows };
    /* Small draws (round 6).  The chip runs 4096 marching waves at a time; a whole image over a small mosaic has too few of them
     * for their lengths to average out - BASELINE's configs[1] (3x3 tiles, 8000 x 2000): 20.6 K waves, 125 us of work per slot,
     * a kernel of 216 us that ends on the far zones' 32-row waves; configs[0] (2000 x 500): 2 K waves, the kernel as long as
     * its longest.  Fewer than 32 K waves: the far zones in 16 rows, the next in 8, two rows to a wave next to the viewer -
     * configs[0] 0.106 -> 0.069 ms per render of a series, 3x3 tiles at 4000 x 1000 0.200 -> 0.188, configs[1] 0.257 -> 0.249
     * (two alternating sweeps over seven settings: profiles/r6_small_images.txt).  Sectors keep their own rules above. */
    if(p.SW == p.W)
    {
        long waves = 0;
        for(int k=0; k<MR_NZONES; k++) waves += (z.row0[k+1] - z.row0[k] + rows[k]-1)/rows[k];
        waves *= (p.N-1 + MR_COLS-1)/MR_COLS;
        if(waves < 32768) { rows[0] = rows[6] = rows[0] < 16 ? rows[0] : 16; rows[1] = rows[5] = rows[1] < 8 ? rows[1] : 8; rows[2] = rows[4] = 2; }
    }
    /* segment numbers (= blockIdx.y = dispatch order) are handed out to the
     * zones with the longest segments first: the long far-field waves start
     * early and the kernel ends on short ones.  With the early depth test
     * (second round of a two-round draw) the order is from the viewer's row
     * outwards instead, so that the ridges in between are in the framebuffer
     * before the far field is tested against it. */
    const int order_far_first [MR_NZONES] = { 0, 6, 1, 5, 2, 4, 3 };
    const int order_near_first[MR_NZONES] = { 3, 2, 4, 1, 5, 0, 6 };
    z.near_first = near_first ? 1 : 0;
    const int* order = near_first ? order_near_first : order_far_first;
    int seg = 0;
    for(int o=0; o<MR_NZONES; o++)
    {
        const int k = order[o];
        z.rows[k] = rows[k];
        z.seg0[k] = seg;
        const int n = z.row0[k+1] - z.row0[k];
        z.nseg[k] = (n + rows[k]-1)/rows[k];
        seg += z.nseg[k];
    }
    z.total = seg;
    return z;
}

hz_params_t hz_make_params(const hz_dev_t* d, const hz_view_t* v)
{
    hz_params_t p;
    memset(&p, 0, sizeof(p));
    p.u.viewer_cell_i  = v->viewer_cell_i;
    p.u.viewer_cell_j  = v->viewer_cell_j;
    p.u.viewer_z       = v->viewer_z;
    p.u.cos_viewer_lat = v->cos_viewer_lat;
    p.u.deg_per_cell   = v->deg_per_cell;
    p.u.aspect         = v->aspect;
    p.u.znear          = v->znear;
    p.u.zfar           = v->zfar;
    p.u.znear_color    = v->znear_color;
    p.u.zfar_color     = v->zfar_color;
    hz_frame_from_az(v->az_deg0, v->az_deg1, &p.u.az_center, &p.u.az_ndc_per_rad);
    p.halfW = (float)d->W * 0.5f;
    p.halfH = (float)d->H * 0.5f;
    p.N = d->N; p.W = d->W; p.H = d->H;
    p.col0 = d->col0; p.col1 = d->col1; p.SW = d->col1 - d->col0;
    p.touched = d->d_touched[d->fbi]; p.seg_stride = d->seg_stride;
    /* Whole panorama on one GPU: the marching waves keep everything up to 64
     * pixels (cheapest in total).  One azimuth sector of several: the waves next
     * to the viewer become the critical path, so medium boxes are handed to
     * k_mid, which spreads them over the chip (measured: 8 sectors 0.97 -> 0.58 ms). */
    p.inline_max = (p.SW == p.W) ? HZ_INLINE_MAX_PIX : 16;
    p.far_dd = (v->zfar*1.001f)*(v->zfar*1.001f);
    {
        /* the mosaic's corners are its most distant vertices */
        const float e0 = hz_abs(hz_east(&p.u, 0.f)),  e1 = hz_abs(hz_east(&p.u, (float)(p.N-1)));
        const float n0 = hz_abs(hz_north(&p.u, 0.f)), n1 = hz_abs(hz_north(&p.u, (float)(p.N-1)));
        const float em = e0 > e1 ? e0 : e1, nm = n0 > n1 ? n0 : n1;
        p.far_strips = !(nm*nm + em*em <= p.far_dd);
    }
    p.big_min    = HZ_INLINE_MAX_PIX;
    p.z_guard = 1.0f/500.0f + (float)(d->W > d->H ? d->W : d->H) * (1.0f/4194304.0f);
    p.z_hide_k = 1.03f * p.z_guard * 16777215.f;
    /* (0 = no cell is ever culled the short way: tiny images, and sides that do not fit the packed 16-bit pixel boxes) */
    p.quad_max_dx = d->W >= 64 && d->W <= 65535 && d->H <= 65535 ? 256*(d->W/16 - 1) : 0;
#ifdef HZ_EXPERIMENTS
    p.exp_fb[HZ_WHO_MARCH] = d->exp.exp_fb_march; p.exp_fb[HZ_WHO_BIG] = d->exp.exp_fb_big;
    p.debug   = d->exp.march_debug;
#endif
    p.pretest_march = 0;                /* (the second round of a two-round draw may switch it on: draw_impl) */
    p.fast_ok = hzf_draw_ok(&p.u) && d->env.fast_math;
    return p;
}

/* ---- which strips can reach the drawn columns -----------------------------------
 * A draw that does not cover the full circle - one GPU's azimuth sector of a
 * panorama, or a view of less than 360 degrees - needs only the strips of the
 * DEM that lie in the wedge of azimuths behind its columns.  Launching every
 * strip and letting the others leave (k_march's corner test) costs a sector the
 * whole grid's launch plus a vertex transform per wave: 0.3 ms of a 0.4 ms
 * sector at 8 sectors.  So the host lists, per draw, the (segment, strip column)
 * pairs worth launching: per segment - a band of rows, i.e. of north offsets -
 * the east extent of wedge x band, in double precision with margins (4 pixels of
 * azimuth, a cell in every direction).  The list only has to be a superset: the
 * corner test stays in the kernel and decides with the rasteriser's own arithmetic. */

/* east extent [lo,hi] of { t*(sin a, cos a) : t >= 0, a in [a0,a1] } intersected with
 * the band n_lo <= n <= n_hi; a1 - a0 <= pi (convex).  false: empty. */
/* sin and cos of a ray's azimuth: the same two or three angles for every band of rows of a walk - kept */
static void ray_sincos(double a, double* s, double* c)
{
    static thread_local double ka[4] = { 1e300, 1e300, 1e300, 1e300 }, ks[4], kc[4];
    static thread_local int turn = 0;
    for(int k=0; k<4; k++) if(ka[k] == a) { *s = ks[k]; *c = kc[k]; return; }
    const int k = turn++ & 3;
    ka[k] = a; ks[k] = sin(a); kc[k] = cos(a);
    *s = ks[k]; *c = kc[k];
}

static bool wedge_band_extent(double a0, double a1, double n_lo, double n_hi, double* lo, double* hi)
{
    const double inf = 1e300;
    double e_lo = inf, e_hi = -inf;
    auto add = [&](double e) { if(e < e_lo) e_lo = e; if(e > e_hi) e_hi = e; };
    if(n_lo <= 0.0 && 0.0 <= n_hi) add(0.0);                   /* the apex */
    const double rays[2] = { a0, a1 };
    for(int r=0; r<2; r++)
    {
        double s, c;
        ray_sincos(rays[r], &s, &c);
        if(fabs(c) < 1e-12)
        {
            if(n_lo <= 0.0 && 0.0 <= n_hi) add(s > 0 ? inf : -inf);
            continue;
        }
        const double bounds[2] = { n_lo, n_hi };
        for(int b=0; b<2; b++)
        {
            const double t = bounds[b]/c;
            if(t >= 0.0) add(t*s);
        }
    }
    if(e_lo > e_hi) return false;
    /* unbounded towards east / west: the wedge contains that direction */
    auto contains = [&](double dir) { double x = fmod(dir - a0, 2.0*M_PI); if(x < 0) x += 2.0*M_PI; return x <= a1 - a0; };
    if(contains( 0.5*M_PI)) e_hi =  inf;
    if(contains(-0.5*M_PI)) e_lo = -inf;
    *lo = e_lo; *hi = e_hi;
    return true;
}

/* strip columns [x0,x1] of the band of cell rows jbeg..jend that can reach the
 * azimuths [a0,a1] (radians, a1 - a0 < 2 pi); false: none */
static bool strips_behind_columns(const hz_params_t& p, double a0, double a1, int jbeg, int jend, int nsx, int* x0, int* x1)
{
    const double m_per_cell_n = (double)HZ_REARTH_PI * (double)p.u.deg_per_cell / 180.0;
    const double m_per_cell_e = m_per_cell_n * (double)p.u.cos_viewer_lat;
    const double n_lo = ((double)(jbeg-1) - (double)p.u.viewer_cell_j) * m_per_cell_n;
    const double n_hi = ((double)(jend+1) - (double)p.u.viewer_cell_j) * m_per_cell_n;
    double lo = 0, hi = 0;
    bool any = false;
    const int parts = (a1 - a0 > M_PI) ? 2 : 1;               /* a wedge of more than 180 degrees: two convex halves */
    for(int k=0; k<parts; k++)
    {
        const double b0 = a0 + (a1 - a0)*k/parts, b1 = a0 + (a1 - a0)*(k+1)/parts;
        double l, h;
        if(!wedge_band_extent(b0, b1, n_lo, n_hi, &l, &h)) continue;
        if(!any) { lo = l; hi = h; any = true; }
        else { if(l < lo) lo = l; if(h > hi) hi = h; }
    }
    if(!any) return false;
    /* cells, then strip columns; a strip reaches MR_COLS cells east of its first column */
    double i_lo = lo/m_per_cell_e + (double)p.u.viewer_cell_i - 2.0, i_hi = hi/m_per_cell_e + (double)p.u.viewer_cell_i + 2.0;
    if(!(i_lo > -1e9)) i_lo = -1e9;
    if(!(i_hi <  1e9)) i_hi =  1e9;
    /* strip sx holds the vertex columns sx*MR_COLS .. sx*MR_COLS + MR_COLS: it meets [i_lo, i_hi] iff sx*MR_COLS <= i_hi and
     * sx*MR_COLS + MR_COLS >= i_lo  (round 6: the western end was floor(i_lo/MR_COLS) - 1, one strip too many in every band) */
    int a = (int)ceil(i_lo/(double)MR_COLS) - 1, b = (int)floor(i_hi/(double)MR_COLS);
    if(a < 0) a = 0;
    if(b > nsx-1) b = nsx-1;
    if(a > b) return false;
    *x0 = a; *x1 = b;
    return true;
}

/* the azimuths behind image columns [col0,col1) +- 4 pixels; false: (nearly) the full circle */
bool hz_azimuths_of_columns(const hz_params_t& p, double* a0, double* a1)
{
    const double k = (double)p.u.az_ndc_per_rad, c = (double)p.u.az_center, hw = (double)p.halfW;
    const double lo = c + (((double)p.col0 - 4.0)/hw - 1.0)/k, hi = c + (((double)p.col1 + 4.0)/hw - 1.0)/k;
    if(!(hi - lo < 2.0*M_PI - 1e-3) || !(hi > lo)) return false;
    *a0 = lo; *a1 = hi;
    return true;
}

/* the (segment, strip column) items of one k_march launch (p.pass says which
 * round's) into `out`, in dispatch order: segments as mr_make_zones numbered
 * them, strip columns west to east */
/* ... the items of BOTH rounds of a two-round draw in one walk over the segments (first: the strips next to the viewer, as
 * k_march's pass 1 decides it; second: all the others), or, one_round, every listed item into `second`.  A viewer that
 * moves brings new lists with every draw - a call into host memory four sectors' worth - and this walk was 0.4 ms of the
 * host's time per sector: the rays' sines and cosines once per call, the per-strip test only where it can fail. */
void hz_list_rounds(const hz_params_t& p, const mr_zones_t& zn, double a0, double a1, bool one_round,
                    std::vector<uint32_t>* first, std::vector<uint32_t>& second, bool every_strip)
{
    const int nsx = (p.N-1 + MR_COLS-1)/MR_COLS;
    if(first) first->clear();
    second.clear();
    /* Round 6: the band's east extent (strips_behind_columns) is the test along the patch's own axes; a patch (a strip's
     * cells in a band of rows: a rectangle in east and north, the viewer outside it) can lie inside that extent and still
     * beside the wedge - in the corner between a ray and the band's edge.  What separates a rectangle from a convex wedge
     * besides its own axes are the wedge's two rays: a patch with all four corners on the outer side of one of them is not
     * listed (two cross products per corner; the patch two cells larger east and west, one north and south).  The patches
     * of a band that pass are neighbours (a convex wedge cuts a convex piece out of the band): the test walks in from both
     * ends of the extent and stops at the first patch that passes.  A wedge wider than 180 degrees is taken in two halves
     * and every patch of the extent tested.  The eight sectors of the benchmark panorama: 1.03 of the grid's waves in sum. */
    const double m_per_cell_n = (double)HZ_REARTH_PI * (double)p.u.deg_per_cell / 180.0;
    const double m_per_cell_e = m_per_cell_n * (double)p.u.cos_viewer_lat;
    const int parts = (a1 - a0 > M_PI) ? 2 : 1;
    double ray[3][2];                               /* (sin, cos) of the parts' edges */
    for(int k=0; k<=parts; k++) { const double a = a0 + (a1 - a0)*k/parts; ray[k][0] = sin(a); ray[k][1] = cos(a); }
    for(int seg=0; seg<zn.total; seg++)
    {
        int jbeg, jend;
        mr_segment_rows(zn, seg, &jbeg, &jend);
        int x0 = 0, x1 = nsx-1;
        if(!every_strip && !strips_behind_columns(p, a0, a1, jbeg, jend, nsx, &x0, &x1)) continue;
        const bool near_rows = jbeg < p.near_j1 && jend > p.near_j0;
        const double n_lo = ((double)(jbeg-1) - (double)p.u.viewer_cell_j) * m_per_cell_n;
        const double n_hi = ((double)(jend+1) - (double)p.u.viewer_cell_j) * m_per_cell_n;
        auto reaches = [&](int sx) -> bool
        {
            const double e_lo = ((double)(sx*MR_COLS - 2) - (double)p.u.viewer_cell_i) * m_per_cell_e;
            const double e_hi = ((double)(sx*MR_COLS + MR_COLS + 2) - (double)p.u.viewer_cell_i) * m_per_cell_e;
            if(e_lo <= 0.0 && 0.0 <= e_hi && n_lo <= 0.0 && 0.0 <= n_hi) return true;      /* (the patch holds the viewer) */
            for(int k=0; k<parts; k++)
            {
                /* cross(ray, corner) = sin*n - cos*e: negative = clockwise of the ray.  Inside the part: clockwise of its
                 * first ray (or on it) and counter-clockwise of its second */
                const double c[4][2] = { { e_lo, n_lo }, { e_hi, n_lo }, { e_lo, n_hi }, { e_hi, n_hi } };
                bool before_first = true, beyond_second = true;
                for(int q=0; q<4; q++)
                {
                    if(!(ray[k][0]*c[q][1] - ray[k][1]*c[q][0] > 0.0))   before_first = false;
                    if(!(ray[k+1][0]*c[q][1] - ray[k+1][1]*c[q][0] < 0.0)) beyond_second = false;
                }
                if(!before_first && !beyond_second) return true;
            }
            return false;
        };
        if(!every_strip && parts == 1)
        {
            while(x0 <= x1 && !reaches(x0)) x0++;
            while(x1 > x0 && !reaches(x1)) x1--;
        }
        for(int sx=x0; sx<=x1; sx++)
        {
            if(!every_strip && parts == 2 && !reaches(sx)) continue;
            const bool near = !one_round && near_rows && sx >= p.near_x0 && sx <= p.near_x1;      /* (as k_march decides it) */
            if(near) { if(first) first->push_back(MR_ITEM(seg, sx)); }
            else second.push_back(MR_ITEM(seg, sx));
        }
    }
}

/* the items of one k_march launch (p.pass: 0 the one round of a one-round draw, 1 / 2 the rounds of a two-round draw) */
void hz_list_items(const hz_params_t& p, const mr_zones_t& zn, double a0, double a1, std::vector<uint32_t>& out, bool every_strip)
{
    if(p.pass == 1) { std::vector<uint32_t> rest; hz_list_rounds(p, zn, a0, a1, false, &out, rest, every_strip); }
    else hz_list_rounds(p, zn, a0, a1, p.pass == 0, NULL, out, every_strip);
}

/* the draw's plan: one round or two, and which strips are "next to the viewer" */
int hz_plan_rounds(const hz_dev_t* d, const hz_view_t* view, hz_params_t& p)
{
    const int nsx = (p.N-1 + MR_COLS-1)/MR_COLS;
    /* The first round's reach: the cells that are wider than ~20 pixels on screen - a cell r rows
     * from the viewer is about ppr/r pixels wide (ppr = pixels per radian of azimuth), so r = ppr/20:
     * 127 cells for a 16000-wide panorama (where 32..256 were timed: hz_k_march.h), 64 for 8000, 260
     * for 32768, at most HZ_NEAR_CELLS_WIDE - and HZ_NEAR_CELLS_MAX for views zoomed far enough (see there).
     * profiles/r3_scenes.json holds the sweep over the scenes of tools/scenes.py.
     * (A middle round between the two - the ring out to 640 cells with the early test against the first round's
     * tables - was built and measured in round 4: two of seven zoomed views gained, five paid its fixed cost, 10.7 ->
     * 11.0 ms in sum; removed in round 5, profiles/r4_middle_round.txt.) */
    int near_cells = d->env.near_cells;
    if(near_cells < 0)
    {
        const float ppr = p.halfW * p.u.az_ndc_per_rad;
        near_cells = (int)(ppr / HZ_NEAR_PX + 0.5f);
        if(near_cells < 16) near_cells = 16;
        if(near_cells > HZ_NEAR_CELLS_WIDE) near_cells = HZ_NEAR_CELLS_WIDE;
        /* (zoomed even at the long reach: there, if the draws before say so - hz_k_march.h, adapt) */
        if(ppr/(float)HZ_NEAR_CELLS_MAX >= HZ_HIZ_MIN_PX && (d->env.adapt == 2 || (d->env.adapt == 1 && d->adapt.long_reach))) near_cells = HZ_NEAR_CELLS_MAX;
    }
    p.near_x0 = (int)floorf((p.u.viewer_cell_i - (float)near_cells)/(float)MR_COLS);
    p.near_x1 = (int)floorf((p.u.viewer_cell_i + (float)near_cells)/(float)MR_COLS);
    if(p.near_x0 < 0) p.near_x0 = 0;
    if(p.near_x1 > nsx-1) p.near_x1 = nsx-1;
    p.near_j0 = (int)floorf(p.u.viewer_cell_j - (float)near_cells);
    p.near_j1 = (int)ceilf (p.u.viewer_cell_j + (float)near_cells);
    /* Two rounds pay where there is terrain behind the first round's strips to be hidden by them
     * and enough pixels for the second round's early depth test to save work; a small image is
     * faster in one round (three kernel launches less).  Measured over the scenes of tools/scenes.py
     * (profiles/r3_scenes.json, ms per render one round / two rounds): 2000x500 0.143 / 0.167,
     * 4000x1000 0.224 / 0.225, 8000x2000 0.353 / 0.316 (a batch of viewpoints of that size 0.512 /
     * 0.465), 16000x4000 1.33 / 1.12, with the API's 40 km far clip 0.752 / 0.725, 32768x8192 12.5 /
     * 11.1 - so: from 6 Mpix on, and a far clip at least three reaches of the first round away.
     * Azimuth sectors decide by the size of the whole image: their renders overlap just the same. */
    const float cells_to_zfar = view->zfar / (p.u.deg_per_cell * 111194.9f);
    const bool want_two = d->env.rounds > 0 ? d->env.rounds == 2
                             : ((double)p.W*(double)p.H >= HZ_TWO_ROUNDS_MIN_PIX && cells_to_zfar >= 3.0f*(float)near_cells);
    return want_two && near_cells > 0 && p.near_x1 >= p.near_x0 ? 2 : 1;
}
