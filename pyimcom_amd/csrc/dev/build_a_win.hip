// build_a_win.hip -- the A builder around LDS-resident table windows (round 2).
//
// PSF-overlap system matrix of a batch of stamps by D5512 table interpolation (reference
// src/pyimcom/psfutil.py:1401-1495 _call_ii_cross, 1597-1732 _call_ii_self, sub-block assembly of
// coadd.py:1027-1068; the 10x10 stencil of routine.py:125-253).
//
// Why windows.  The first builder (build_a.hip, kept as IMCOM_BUILD_A=legacy) DMAs every sample's own 10 x 96 B
// stencil rows: 960 B per sample from L2 at the ~30 B/clk/CU the L2 -> CU path sustains = 32 clk per sample and
// CU, the 132 us per cfg-2 stamp it measures.  But the pixels of one (InStamp, exposure) piece are a compact patch
// of a detector lattice, so the separations (x_i - x_j, y_i - y_j) of a block of rows against a block of columns
// of ONE exposure pair fill a small rectangle of ONE table: its bounding box (+ the 10 taps) is staged once
// (~200 B per sample) and the taps are read from LDS.
//
// Work decomposition.  One 512-thread workgroup per CU = one index-aligned block of A: 32 rows x 64 columns, blocks
// on or above the diagonal.  The rows and the columns split into pieces of equal stamp-local PSF index (a
// handful); every (row piece, column piece) rectangle has one pair code, i.e. one table.
//   plan     all rectangles at once, 16 lanes per rectangle: the column piece is offered whole, in halves, quarters
//            and eighths; the bounding box of a candidate's cells follows from the min / max of x and y over its row
//            range and its column range (the separations of a rectangle of samples fill exactly the Minkowski
//            difference of the two ranges; +-1 cell of slack because the plan multiplies by 1/dscale where the
//            samples divide); the coarsest candidates whose window fits become plan entries; eighths that do not
//            fit are tried against the two halves of the row piece, and what is left is interpolated straight
//            from global memory (correct for any input, only slow; so is a block with more pieces than the plan holds);
//   execute  entry k+1 is staged by LDS-DMA into one window buffer while entry k is evaluated from the other: one
//            barrier per entry, the L2 -> LDS stream never pauses.  A sample reads its 10 x 12 doubles with
//            ds_read_b128 and accumulates in the reference's order (inner sum over x taps, outer over y taps; a
//            flipped table is read back to front);
//   store    the 32 x 64 results sit in an LDS tile and leave as whole 256 / 512-byte row segments, the block and
//            its mirror image, so A is exactly symmetric and every element is written exactly once.  Rows and
//            columns beyond n[s] get the identity out to ldn.
//
// LDS image of a window.  Detector lattices make consecutive samples step by `oversamp` = 8 table samples = 4
// 16-byte units; in a plain row-major image the 16 lanes of a ds_read_b128 group would hit only four bank groups.
// One pad slot after every four units turns the step into 5 slots = 20 banks, which walks all 64 banks in 16 lanes.
// The DMA image stays lane-linear (slot g of the window <- lane g); lanes that land on pad slots are masked off.
#include <cstdlib>

#include "common.h"
#include "d5512.h"
#include "launchers.h"

namespace imcom {

namespace {

constexpr int PAIR_SWAP = 1 << 29, PAIR_FLIP = 1 << 30, PAIR_MASK = (1 << 28) - 1;
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int TR = 32, TC = 64, NT = 256;
constexpr int TP = TC + 1;        // pitch of the result tile (doubles): column reads hit 32 distinct bank pairs
constexpr int WIN_SLOTS = 3392;   // 16-byte slots of the window (53 KB); two workgroups per CU
constexpr int RPMAX = 6, CPMAX = 8, PLAN_MAX = 64;
constexpr int CNODES = 15;        // a column piece whole, in halves, quarters, eighths
constexpr int RNODES = 3;         // a row piece whole and in halves

#define IMCOM_GLDS16(gptr, ldsptr)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),           \
                                     (__attribute__((address_space(3))) void *)(ldsptr), 16, 0, 0)

// linear index over the blocks on or above the diagonal: block row bi holds the block columns bi/2 .. ncb-1
// (TC = 2 TR); the rows 2k and 2k+1 hold ncb - k blocks each.
__device__ __forceinline__ void block_index(long t, int ncb, int &bi, int &bj)
{
    // blocks before the row pair k: 2 k ncb - k (k - 1)
    const double b = 2.0 * ncb + 1.0;
    int k = (int)((b - sqrt(b * b - 4.0 * (double)t)) * 0.5);
    if (k < 0) k = 0;
    while (k > 0 && 2L * k * ncb - (long)k * (k - 1) > t) k--;
    while (2L * (k + 1) * ncb - (long)(k + 1) * k <= t) k++;
    const long r = t - (2L * k * ncb - (long)k * (k - 1));
    const int per = ncb - k;
    bi = 2 * k + (r >= per ? 1 : 0);
    bj = k + (int)(r >= per ? r - per : r);
}

// the reference's coordinate arithmetic (psfutil.py:1683-1690): difference, divide, add nc, add the 6-sample border
__device__ __forceinline__ void sep(double xa, double ya, double xb, double yb, double dscale, double nc, double &dx, double &dy)
{
    dx = xa - xb; dy = ya - yb;
    dx /= dscale; dx += nc; dx += 6.0;
    dy /= dscale; dy += nc; dy += 6.0;
}

struct PlanEntry {
    short r_lo, r_hi, c_lo, c_hi;  // rows / columns of the block, local indices
    int ox0, oy0, H, sp;           // window: first table column / row, rows, slots per row
    int code;                      // pair code (< 0: no table, the samples are the penalty alone)
    int mode;                      // 0 window, 1 straight from global memory, 2 nothing on the table
    double pen;
};

// [lo, hi) of node `n` of a range [a, b): n = 0 whole, 1-2 halves, 3-6 quarters, 7-14 eighths
__device__ __forceinline__ void node_range(int a, int b, int n, int &lo, int &hi)
{
    const int lev = n >= 7 ? 3 : (n >= 3 ? 2 : (n >= 1 ? 1 : 0));
    const int k = n - ((1 << lev) - 1), len = b - a;
    lo = a + ((k * len) >> lev);
    hi = a + (((k + 1) * len) >> lev);
}

struct Window {
    int ox0, oy0, H, sp;
    bool any, fits;
};

// window of the samples {row range} x {column range} from the extremes rm = (xmin, xmax, ymin, ymax) of the rows
// and cm of the columns
__device__ __forceinline__ Window window_of(const double *rm, const double *cm, bool swap, bool rev, int ng, double nc,
                                            double inv_dscale)
{
    // separations a - b with a = row pixel, b = column pixel (the other way round for SWAP)
    double dxlo = rm[0] - cm[1], dxhi = rm[1] - cm[0], dylo = rm[2] - cm[3], dyhi = rm[3] - cm[2];
    if (swap) { const double t0 = -dxhi, t1 = -dyhi; dxhi = -dxlo; dyhi = -dylo; dxlo = t0; dylo = t1; }
    const double off = nc + 6.0;
    int cx0 = to_cell(dxlo * inv_dscale + off) - 1, cx1 = to_cell(dxhi * inv_dscale + off) + 1;
    int cy0 = to_cell(dylo * inv_dscale + off) - 1, cy1 = to_cell(dyhi * inv_dscale + off) + 1;
    cx0 = max(cx0, 4); cy0 = max(cy0, 4); cx1 = min(cx1, ng - 6); cy1 = min(cy1, ng - 6);
    Window w;
    w.any = cx0 <= cx1 && cy0 <= cy1;
    w.fits = false;
    w.ox0 = w.oy0 = w.H = w.sp = 0;
    if (w.any) {
        // first table column / row of a sample's 10 x 10 taps, in the coordinates of the stored table
        w.ox0 = rev ? ng - cx1 - 6 : cx0 - 4;
        w.oy0 = rev ? ng - cy1 - 6 : cy0 - 4;
        const int W = cx1 - cx0 + 10;
        w.H = cy1 - cy0 + 10;
        const int upr = (W + 3) >> 1;   // units per row: the 16-byte alignment shifts a row by one double, and the 12
                                        // doubles a sample reads may end one double beyond its taps
        w.sp = upr + ((upr + 3) >> 2);  // + one pad slot after every four units
        w.fits = (long)w.H * w.sp <= WIN_SLOTS;
    }
    return w;
}

}  // namespace

__global__ __launch_bounds__(NT, 2) void build_A_win_kernel(const int *__restrict__ n, int ldn, const double *__restrict__ x,
                                                            const double *__restrict__ y, const int *__restrict__ psf,
                                                            const double *__restrict__ tables, long tab_elems, int ng, double nc,
                                                            double dscale, const int *__restrict__ pair_tab,
                                                            const double *__restrict__ pair_pen, int npsf_max,
                                                            double *__restrict__ A, int ncb, long nblocks, int dbg)
{
    __shared__ __attribute__((aligned(16))) f64x2 winA[WIN_SLOTS];
    __shared__ double tile[TR][TP];
    __shared__ double xr[TR], yr[TR], xc[TC], yc[TC];
    __shared__ int pr[TR], pc[TC];
    __shared__ unsigned long long masks[2];
    __shared__ short rpl[RPMAX + 1], cpl[CPMAX + 1];
    __shared__ double rnode[RPMAX * RNODES][4], cnode[CPMAX * CNODES][4];
    __shared__ PlanEntry plan[PLAN_MAX];
    __shared__ int nplan;
    const int s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: every XCD walks one contiguous eighth
    // of the block list, so that the blocks it works on at one time use the same few tables (gridDim.x is the block
    // count padded to a multiple of 8).
    const long per = (nblocks + 7) / 8;
    const long t = (((long)blockIdx.y * gridDim.x + blockIdx.x) & 7) * per + ((long)blockIdx.x >> 3);
    if (t >= nblocks) return;
    int bi, bj;
    block_index(t, ncb, bi, bj);
    const int r0 = bi * TR, c0 = bj * TC;
    if (r0 >= ldn) return;
    const int ns = n[s];
    const long base = (long)s * ldn;
    double *As = A + (long)s * ldn * ldn;
    const double inv_dscale = 1.0 / dscale;
    const bool square = c0 < r0 + TR;  // the block holds the diagonal square rows x rows (c0 <= r0 < c0 + TC)

    // ---- rows, columns, their pieces
    if (tid < TR) {
        const int i = r0 + tid;
        const bool ok = i < ns;
        xr[tid] = ok ? x[base + i] : 0.0;
        yr[tid] = ok ? y[base + i] : 0.0;
        pr[tid] = ok ? psf[base + i] : -1;
    } else if (tid >= 64 && tid < 64 + TC) {
        const int q = tid - 64, j = c0 + q;
        const bool ok = j < ns;
        xc[q] = ok ? x[base + j] : 0.0;
        yc[q] = ok ? y[base + j] : 0.0;
        pc[q] = ok ? psf[base + j] : -1;
    }
    if (tid == 0) nplan = 0;
    // identity everywhere first: padding rows / columns keep it, samples overwrite it
    for (int e = tid; e < TR * TC; e += NT) {
        const int li = e / TC, lj = e % TC;
        tile[li][lj] = (r0 + li == c0 + lj) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (wave == 0) {
        const bool first = lane < TR && (lane == 0 || pr[lane] != pr[lane - 1]);
        const unsigned long long m = __ballot(first);
        if (lane == 0) masks[0] = m;
        if (first) { const int k = __popcll(m & ((1ULL << lane) - 1)); if (k < RPMAX) rpl[k] = (short)lane; }
        const int np = __popcll(m);
        if (lane == 0 && np <= RPMAX) rpl[np] = TR;
    } else if (wave == 1) {
        const bool first = lane == 0 || pc[lane] != pc[lane - 1];
        const unsigned long long m = __ballot(first);
        if (lane == 0) masks[1] = m;
        if (first) { const int k = __popcll(m & ((1ULL << lane) - 1)); if (k < CPMAX) cpl[k] = (short)lane; }
        const int np = __popcll(m);
        if (lane == 0 && np <= CPMAX) cpl[np] = TC;
    }
    __syncthreads();
    const int nrp = __popcll(masks[0]), ncp = __popcll(masks[1]);
    // more pieces than the plan holds, or a table with an even row pitch (the window reader counts on alternating row
    // parities): every sample straight from global memory
    bool whole_direct = nrp > RPMAX || ncp > CPMAX || !(ng & 1);

    if (!whole_direct) {
        // ---- extremes of every candidate range
        if (tid < ncp * CNODES) {
            const int q = tid / CNODES, nd = tid - q * CNODES;
            int lo, hi;
            node_range(cpl[q], cpl[q + 1], nd, lo, hi);
            double a0 = 1e300, a1 = -1e300, b0 = 1e300, b1 = -1e300;
            for (int k = lo; k < hi; k++) { a0 = fmin(a0, xc[k]); a1 = fmax(a1, xc[k]); b0 = fmin(b0, yc[k]); b1 = fmax(b1, yc[k]); }
            cnode[tid][0] = a0; cnode[tid][1] = a1; cnode[tid][2] = b0; cnode[tid][3] = b1;
        } else if (tid >= 192 && tid - 192 < nrp * RNODES) {
            const int u = tid - 192, p = u / RNODES, nd = u - p * RNODES;
            int lo, hi;
            node_range(rpl[p], rpl[p + 1], nd, lo, hi);
            double a0 = 1e300, a1 = -1e300, b0 = 1e300, b1 = -1e300;
            for (int k = lo; k < hi; k++) { a0 = fmin(a0, xr[k]); a1 = fmax(a1, xr[k]); b0 = fmin(b0, yr[k]); b1 = fmax(b1, yr[k]); }
            rnode[u][0] = a0; rnode[u][1] = a1; rnode[u][2] = b0; rnode[u][3] = b1;
        }
        __syncthreads();
        // ---- the plan: one rectangle (row piece, column piece) per group of 16 lanes
        const int grp = tid >> 4, gl = tid & 15, npairs = nrp * ncp;
        for (int pp0 = 0; pp0 < npairs; pp0 += NT / 16) {
            const int pp = pp0 + grp;
            const bool live = pp < npairs;
            const int p = live ? pp / ncp : 0, q = live ? pp - p * ncp : 0;
            const int ra = rpl[p], rb = rpl[p + 1], ca = cpl[q], cb = cpl[q + 1];
            const int pi = pr[ra], pj = pc[ca];
            // skipped: padding rows / columns, and rectangles wholly below the diagonal (the mirror image of another block)
            const bool work = live && pi >= 0 && pj >= 0 && !(c0 + cb - 1 < r0 + ra);
            long pidx = 0;
            int code = -1;
            double pen = 0.0;
            if (work) { pidx = ((long)s * npsf_max + pi) * npsf_max + pj; code = pair_tab[pidx]; pen = pair_pen[pidx]; }
            const bool has = code >= 0, swap = has && (code & PAIR_SWAP), rev = has && (code & PAIR_FLIP);
            auto emit = [&](int rlo, int rhi, int clo, int chi, const Window &w, int mode) {
                const int idx = atomicAdd(&nplan, 1);
                if (idx < PLAN_MAX) {
                    PlanEntry e;
                    e.r_lo = (short)rlo; e.r_hi = (short)rhi; e.c_lo = (short)clo; e.c_hi = (short)chi;
                    e.ox0 = w.ox0; e.oy0 = w.oy0; e.H = w.H; e.sp = w.sp; e.code = code; e.mode = mode; e.pen = pen;
                    plan[idx] = e;
                }
            };
            // round 1: the whole row piece against the 15 candidates of the column piece
            int lo = 0, hi = 0;
            Window w{};
            bool cand = false;
            if (work && has && gl < CNODES) {
                node_range(ca, cb, gl, lo, hi);
                cand = hi > lo;
                if (cand) w = window_of(rnode[p * RNODES], cnode[q * CNODES + gl], swap, rev, ng, nc, inv_dscale);
            }
            const bool good = cand && (w.fits || !w.any);  // nothing on the table: no window needed
            const unsigned fm = (unsigned)(__ballot(good) >> (16 * ((tid >> 4) & 3))) & 0xFFFFu;
            bool anc = false;  // an ancestor is good: this candidate is covered by it
            for (int a = gl; a > 0;) { a = (a - 1) >> 1; anc = anc || ((fm >> a) & 1u); }
            if (cand && good && !anc) emit(ra, rb, lo, hi, w, w.any ? 0 : 2);
            if (work && !has && gl == 0) { Window z{}; emit(ra, rb, ca, cb, z, 2); }
            // round 2: eighths that do not fit, against the two halves of the row piece
            const bool leaf_open = cand && gl >= 7 && !good && !anc;
            const unsigned om = (unsigned)(__ballot(leaf_open) >> (16 * ((tid >> 4) & 3))) & 0xFFFFu;
            {
                const int leaf = 7 + (gl & 7), half = 1 + (gl >> 3);
                if (work && has && ((om >> leaf) & 1u)) {
                    int rlo, rhi, clo, chi;
                    node_range(ra, rb, half, rlo, rhi);
                    node_range(ca, cb, leaf, clo, chi);
                    if (rhi > rlo) {
                        const Window w2 = window_of(rnode[p * RNODES + half], cnode[q * CNODES + leaf], swap, rev, ng, nc, inv_dscale);
                        emit(rlo, rhi, clo, chi, w2, w2.any ? (w2.fits ? 0 : 1) : 2);
                    }
                }
            }
        }
        __syncthreads();
        whole_direct = nplan > PLAN_MAX;
    }

    // one sample of the block (rows / columns are local indices): its cell and weights; false = off the table
    auto sample_geom = [&](int li, int lj, int code, int &cx, int &cy, double (&wx)[10], double (&wy)[10]) -> bool {
        const bool swap = code & PAIR_SWAP;
        double dx, dy;
        if (!swap) sep(xr[li], yr[li], xc[lj], yc[lj], dscale, nc, dx, dy);
        else sep(xc[lj], yc[lj], xr[li], yr[li], dscale, nc, dx, dy);
        cx = to_cell(dx); cy = to_cell(dy);
        if (cx < 4 || cx >= ng - 5 || cy < 4 || cy >= ng - 5) return false;
        d5512_getw(wx, dx - cx - 0.5);
        d5512_getw(wy, dy - cy - 0.5);
        return true;
    };
    // straight from global memory
    auto sample_direct = [&](int li, int lj, int code) -> double {
        int cx, cy;
        double wx[10], wy[10];
        if (!sample_geom(li, lj, code, cx, cy, wx, wy)) return 0.0;
        const long t0 = (long)(code & PAIR_MASK) * ng * ng, off = (long)(cy - 4) * ng + (cx - 4);
        return (code & PAIR_FLIP) ? stencil(tables + t0 + ((long)ng * ng - 1 - off), -(long)ng, -1, wx, wy)
                                  : stencil(tables + t0 + off, ng, 1, wx, wy);
    };
    // from a staged window (an LDS array, passed by reference so that the reads stay ds_read_b128)
    auto sample_win = [&](int li, int lj, int code, const f64x2 (&win_)[WIN_SLOTS], int ox0, int oy0, int sp) -> double {
        int cx, cy;
        double wx[10], wy[10];
        if (!sample_geom(li, lj, code, cx, cy, wx, wy)) return 0.0;
        if (dbg & 2) return wx[3] * wy[4];
        const bool rev = code & PAIR_FLIP;
        const long t0 = (long)(code & PAIR_MASK) * ng * ng;
        if (rev) {  // taps of a flipped table in ascending address order: x weights back to front
#pragma unroll
            for (int c = 0; c < 5; c++) { const double tw = wx[c]; wx[c] = wx[9 - c]; wx[9 - c] = tw; }
        }
        // window coordinates of the sample's first tap column / row in the stored table
        const int px = (rev ? ng - cx - 6 : cx - 4) - ox0, py = (rev ? ng - cy - 6 : cy - 4) - oy0;
        // stencil row r sits in window row py + r (plain) or py + 9 - r (flipped).  A window row starts on a 16-byte
        // boundary of the table, i.e. one double early when its first element has an odd index, and that parity
        // alternates from row to row (ng is odd).  So five of a sample's rows have their ten taps in five whole units
        // (type E), the other five in the inner ten of six units (type O); which rows are which depends on the lane.
        const int wr0 = rev ? py + 9 : py, rstep = rev ? -1 : 1;
        const int e0par = (int)((t0 + (long)oy0 * ng + ox0) & 1);
        const int par0 = (e0par ^ (wr0 & 1)) & 1;
        const int f = (px + par0) & 1;                 // E rows are r = 2k + f, O rows r = 2k + 1 - f
        const int uE = (px + (par0 ^ f)) >> 1, uO = (px + (par0 ^ f ^ 1)) >> 1;
        int aE[5], aO[6];
#pragma unroll
        for (int u = 0; u < 5; u++) aE[u] = (uE + u) + ((uE + u) >> 2);
#pragma unroll
        for (int u = 0; u < 6; u++) aO[u] = (uO + u) + ((uO + u) >> 2);
        const int rinc = rstep * sp;
        int rowE = (wr0 + f * rstep) * sp, rowO = (wr0 + (1 - f) * rstep) * sp;
        double val = 0.0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            f64x2 ce[5], co[6];
#pragma unroll
            for (int u = 0; u < 5; u++) ce[u] = win_[rowE + aE[u]];
#pragma unroll
            for (int u = 0; u < 6; u++) co[u] = win_[rowO + aO[u]];
            double se = 0.0, so = 0.0;
#pragma unroll
            for (int m = 0; m < 10; m++) { se += wx[m] * ce[m >> 1][m & 1]; so += wx[m] * co[(m + 1) >> 1][(m + 1) & 1]; }
            const double wye = f ? wy[2 * k + 1] : wy[2 * k], wyo = f ? wy[2 * k] : wy[2 * k + 1];
            val += se * wye;
            val += so * wyo;
            rowE += 2 * rinc;
            rowO += 2 * rinc;
        }
        return val;
    };

    if (whole_direct) {
        for (int e = tid; e < TR * TC; e += NT) {
            const int li = e / TC, lj = e % TC;
            if (pr[li] < 0 || pc[lj] < 0 || c0 + lj < r0 + li) continue;
            const long pidx = ((long)s * npsf_max + pr[li]) * npsf_max + pc[lj];
            const int code = pair_tab[pidx];
            tile[li][lj] = (code >= 0 ? sample_direct(li, lj, code) : 0.0) + pair_pen[pidx];
        }
    } else {
        const int np = nplan;
        auto stage = [&](const PlanEntry &e, f64x2 (&buf)[WIN_SLOTS]) {
            if (e.mode != 0 || (dbg & 1)) return;
            const long t0 = (long)(e.code & PAIR_MASK) * ng * ng;
            const int total = e.H * e.sp, sp = e.sp;
            const float inv_sp = 1.0f / (float)sp;
            for (int g0 = wave * 64; g0 < total; g0 += NT) {
                const int g = g0 + lane;
                int row = (int)((float)g * inv_sp);
                if (row * sp > g) row--;
                if ((row + 1) * sp <= g) row++;
                const int sl = g - row * sp, q5 = sl / 5;
                const bool pad = sl - q5 * 5 == 4;
                const int u = sl - q5;
                const long e0 = t0 + (long)(e.oy0 + row) * ng + e.ox0;
                const long a = (e0 & ~1L) + 2 * u;
                f64x2 *dst = &buf[g0];  // wave-uniform; lane l lands in slot g0 + l
                if (g < total && !pad) {
                    if (a + 2 <= tab_elems) IMCOM_GLDS16(tables + a, dst);
                    else { f64x2 v; v[0] = a < tab_elems ? tables[a] : 0.0; v[1] = 0.0; buf[g] = v; }
                }
            }
        };
        auto evaluate = [&](const PlanEntry &e, const f64x2 (&wbuf)[WIN_SLOTS]) {
            const int ncol = e.c_hi - e.c_lo, cnt = (e.r_hi - e.r_lo) * ncol;
            const float inv_nc = 1.0f / (float)ncol;
            for (int q = tid; q < cnt; q += NT) {
                int li = (int)((float)q * inv_nc);
                if (li * ncol > q) li--;
                if ((li + 1) * ncol <= q) li++;
                const int lj = e.c_lo + (q - li * ncol);
                li += e.r_lo;
                if (c0 + lj < r0 + li) continue;  // below the diagonal: mirrored from (j, i)
                double v = 0.0;
                if (e.mode == 0) v = sample_win(li, lj, e.code, wbuf, e.ox0, e.oy0, e.sp);
                else if (e.mode == 1) v = sample_direct(li, lj, e.code);
                tile[li][lj] = v + e.pen;
            }
        };
        for (int k = 0; k < np; k++) {
            __syncthreads();  // everybody is done with the window (entry k - 1)
            stage(plan[k], winA);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // entry k has landed (the other workgroup of this CU computes meanwhile)
            evaluate(plan[k], winA);
        }
    }
    __syncthreads();

    // ---- write the block and its mirror image as whole row segments
    for (int e = tid; e < TR * TC; e += NT) {
        const int li = e / TC, lj = e % TC;
        const int i = r0 + li, j = c0 + lj;
        if (i >= ldn || j >= ldn) continue;
        if (square && j < r0) continue;  // left of the diagonal square: written by the mirror of another block
        double v;
        if (square && j < i) v = tile[j - r0][i - c0];  // inside the square, below the diagonal
        else v = tile[li][lj];
        As[(long)i * ldn + j] = v;
    }
    for (int e = tid; e < TC * TR; e += NT) {
        const int lj = e / TR, li = e % TR;
        const int i = r0 + li, j = c0 + lj;  // element (i, j) of the block goes to (j, i)
        if (i >= ldn || j >= ldn) continue;
        if (j < r0 + TR) continue;  // rows of the diagonal square (and left of it) are not mirrored from here
        As[(long)j * ldn + i] = tile[li][lj];
    }
}

int launch_build_A_win(imcom_ctx *ctx, int batch, const int *n_dev, int ldn, const double *x, const double *y, const int *psf,
                       const double *tables, int ntab, int ng, double nc, double dscale, const int *pair_tab, const double *pair_pen,
                       int npsf_max, double *A)
{
    IMCOM_REQUIRE((long)ntab * ng * ng < (1L << 40) && ntab <= PAIR_MASK, "table stack too large (%d tables of %d^2)", ntab, ng);
    const int ncb = (ldn + TC - 1) / TC;
    long nblocks = 0;
    for (int k = 0; k < ncb; k++) nblocks += 2L * (ncb - k);
    const long ngrid = (nblocks + 7) / 8 * 8;
    IMCOM_REQUIRE(ngrid < (1L << 31), "A too large for one launch");
    static const int dbg = getenv("IMCOM_A_WIN_DBG") ? atoi(getenv("IMCOM_A_DBG")) : 0;
    hipLaunchKernelGGL(build_A_win_kernel, dim3((unsigned)ngrid, batch), dim3(NT), 0, ctx->stream, n_dev, ldn, x, y, psf, tables,
                       (long)ntab * ng * ng, ng, nc, dscale, pair_tab, pair_pen, npsf_max, A, ncb, nblocks, dbg);
    return check_launch("build_A_win_kernel");
}

}  // namespace imcom
