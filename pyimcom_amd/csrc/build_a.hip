// build_a.hip -- the A builder: PSF-overlap system matrix of a batch of stamps by D5512 table
// interpolation (reference src/pyimcom/psfutil.py:1401-1495 _call_ii_cross, 1597-1732 _call_ii_self, and
// the sub-block assembly of coadd.py:1027-1068).
//
// The reference interpolates the element (i, j) with i before j in the stamp's pixel order and mirrors
// it; so does this kernel: 16x16 sample tiles over the upper triangle of tiles, one sample per thread,
// both the tile and its mirror written row-wise through an LDS transpose.
// pair code: bits 0..27 table index; bit 29 SWAP (the reference evaluated the block from the other
// stamp's side and transposed it, psfutil.py:1990-1996); bit 30 FLIP (np.flip'ed table, 1658-1665).
//
// This is gather-bound work (100 table reads per sample, 2.4 M samples per cfg-2 stamp), no MFMA.
#include <cstdlib>

#include "common.h"
#include "d5512.h"
#include "launchers.h"

namespace imcom {

constexpr int PAIR_SWAP = 1 << 29, PAIR_FLIP = 1 << 30, PAIR_MASK = (1 << 28) - 1;
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void tile_index(long t, int ntile, int &ti, int &tj)
{
    // linear index over the upper triangle of tiles: t = ti*ntile - ti(ti-1)/2 + (tj - ti)
    const double b = 2.0 * ntile + 1.0;
    ti = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    while ((long)ti * ntile - (long)ti * (ti - 1) / 2 > t) ti--;
    while ((long)(ti + 1) * ntile - (long)(ti + 1) * ti / 2 <= t) ti++;
    tj = ti + (int)(t - ((long)ti * ntile - (long)ti * (ti - 1) / 2));
}

// Staging.  For every stencil row the 256 samples of a tile need 256 segments of 10 contiguous doubles
// at arbitrary (8-byte aligned) table positions.  Six consecutive lanes fetch the six 16-byte aligned
// chunks that cover one segment at either parity -- an instruction touches ~11 segments instead of 64
// scattered ones -- and the chunks go global -> LDS directly (LDS-DMA, global_load_lds_dwordx4: the LDS
// image is lane-linear, [sample][6 chunks], exactly the order the lanes are assigned in).  Row r+1 is in
// flight while row r is consumed from the other LDS buffer.
constexpr int SEG_W = 10;  // doubles staged per stencil row of a sample
constexpr int NCH = 5;     // 16-byte chunks per stencil row

#define IMCOM_GLDS16(gptr, ldsptr)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),           \
                                     (__attribute__((address_space(3))) void *)(ldsptr), 16, 0, 0)

// Row buffers per wave; A_RING - 1 rows are in flight.  Three buffers (two workgroups per CU instead of three)
// measured slower, 40.9 against 33.7 ms per 256 cfg-2 stamps: the gather needs the waves more than the depth.
#ifndef IMCOM_A_RING
#define IMCOM_A_RING 2
#endif
constexpr int A_RING = IMCOM_A_RING;
#ifndef IMCOM_A_PMAX
#define IMCOM_A_PMAX 8
#endif
constexpr int A_PMAX = IMCOM_A_PMAX;  // stamp-local PSFs up to which the pair table is staged in LDS (one PSF group: 6 x 6 entries); a stamp of four
                                      // groups (24 x 24) has every thread fetch its own entry instead: staging 576 entries per tile cost 1.3 ms per block

__global__ __launch_bounds__(256, A_RING == 2 ? 3 : 2) void build_A_kernel(const int *__restrict__ n, int ldn,
                                                      const double *__restrict__ x,
                                                      const double *__restrict__ y,
                                                      const int *__restrict__ psf,
                                                      const double *__restrict__ tables, long tab_elems,
                                                      int ng, double nc, double dscale,
                                                      const int *__restrict__ pair_tab,
                                                      const double *__restrict__ pair_pen, int npsf_max,
                                                      double *__restrict__ A, int ntile, const int *__restrict__ desc)
{
    __shared__ __attribute__((aligned(16))) double seg[A_RING][256 * SEG_W];
    __shared__ long seg0[256];  // first element (ascending address order) of stencil row 0, as a 64-bit offset into the table stack
                                // (a block of the reference's size keeps > 2^31 table elements resident); < 0: no loads
    __shared__ int rstep[256];  // element step between stencil rows (+ng, or -ng for a flipped table)
    __shared__ double tile[16][17];
    // the tile's 16 + 16 pixels and the stamp's pair table, fetched once by a few lanes in ONE round trip (every thread doing its
    // own psf -> pair_tab -> pair_pen chain of dependent global loads made the prologue 40 % of this kernel)
    __shared__ int pix_psf[32];
    __shared__ double pix_x[32], pix_y[32];
    __shared__ int ptab[A_PMAX * A_PMAX];
    __shared__ double ppen[A_PMAX * A_PMAX];
    const int s = blockIdx.y, tid = threadIdx.x;
    int ti, tj;
    // Workgroups are dealt round-robin over the 8 XCDs (dispatch order % 8), each with its own L2: give every XCD a
    // contiguous eighth of the tile list, so that the tiles it works on at one time use the same few tables.
    // (gridDim.x is the tile count padded to a multiple of 8.)
    const long ntri = (long)ntile * (ntile + 1) / 2, per = (ntri + 7) / 8;
    long t = (((long)blockIdx.y * gridDim.x + blockIdx.x) & 7) * per + ((long)blockIdx.x >> 3);
    if (t >= ntri) return;
    // desc (optional): this stamp's tiles ordered by the PSF pair of their first sample, so that the tiles an XCD works on at one
    // time read the same table (the stamps of a block spread their samples over ~300 tables of four PSF groups)
    if (desc) t = desc[(long)s * ntri + t];
    tile_index(t, ntile, ti, tj);
    const int ns = n[s];
    const int nfull = (ns + NB - 1) / NB * NB;  // rows/cols the factorisation will ever read
    const int li = tid >> 4, lj = tid & 15;
    const int i = ti * 16 + li, j = tj * 16 + lj;
    double *As = A + (long)s * ldn * ldn;
    if (ti * 16 >= nfull || tj * 16 >= nfull) {
        // beyond this stamp's pixels (a shorter stamp of a ragged batch): identity out to ldn, as imcom_build_A promises
        if (i < ldn && j < ldn) As[(long)i * ldn + j] = (i == j) ? 1.0 : 0.0;
        const int mi = tj * 16 + li, mj = ti * 16 + lj;
        if (ti != tj && mi < ldn && mj < ldn) As[(long)mi * ldn + mj] = 0.0;
        return;
    }
    const long base = (long)s * ldn;
    const bool ptab_lds = npsf_max <= A_PMAX;
    if (tid < 32) {
        const int p = (tid < 16 ? ti : tj) * 16 + (tid & 15);
        const bool in = p < ns;
        pix_psf[tid] = in ? psf[base + p] : 0;
        pix_x[tid] = in ? x[base + p] : 0.0;
        pix_y[tid] = in ? y[base + p] : 0.0;
    }
    if (ptab_lds)
        for (int e = tid; e < npsf_max * npsf_max; e += 256) {
            ptab[e] = pair_tab[(long)s * npsf_max * npsf_max + e];
            ppen[e] = pair_pen[(long)s * npsf_max * npsf_max + e];
        }
    __syncthreads();
    double wx[10], wy[10];
#pragma unroll
    for (int k = 0; k < 10; k++) wx[k] = wy[k] = 0.0;  // samples without a stencil (off the table, lower triangle) add nothing
    double val = 0.0, pen = 0.0;
    bool active = false, rev = false;
    long my_seg0 = -1;
    int my_step = 0;
    double fx = 0.0, fy = 0.0;  // fractional cell positions: the weights are formed after the first stencil row is on its way
    if (i < ns && j < ns && j >= i) {
        const int pe = pix_psf[li] * npsf_max + pix_psf[16 + lj];
        const int code = ptab_lds ? ptab[pe] : pair_tab[(long)s * npsf_max * npsf_max + pe];
        pen = ptab_lds ? ppen[pe] : pair_pen[(long)s * npsf_max * npsf_max + pe];
        if (code >= 0) {
            const bool swap = code & PAIR_SWAP;
            rev = code & PAIR_FLIP;
            const int tab = code & PAIR_MASK;
            double dx = swap ? pix_x[16 + lj] - pix_x[li] : pix_x[li] - pix_x[16 + lj];
            double dy = swap ? pix_y[16 + lj] - pix_y[li] : pix_y[li] - pix_y[16 + lj];
            dx /= dscale; dx += nc; dx += 6.0;
            dy /= dscale; dy += nc; dy += 6.0;
            const int xi = to_cell(dx), yi = to_cell(dy);
            if (!(xi < 4 || xi >= ng - 5 || yi < 4 || yi >= ng - 5)) {
                fx = dx - xi - 0.5;
                fy = dy - yi - 0.5;
                const long t0 = (long)tab * ng * ng, off = (long)(yi - 4) * ng + (xi - 4);
                // flipped table: tap (r, c) is element last - (off + r*ng + c): row r is the ascending
                // segment that starts at last - off - r*ng - 9, read back to front
                my_seg0 = rev ? t0 + ((long)ng * ng - 1 - off - 9) : t0 + off;
                my_step = rev ? -ng : ng;
                active = true;
            }
        }
    }
    seg0[tid] = my_seg0;
    rstep[tid] = my_step;
    double wt[10];  // the x weights as the row loop uses them (set below, once the first row is on its way)
    // A stencil row is staged as the FIVE 16-byte chunks that hold exactly its 10 taps: the LDS-DMA takes 8-byte aligned
    // global addresses, so a chunk may start on any double (round 1 fetched six aligned chunks per row and carried two
    // zero-weight doubles: 96 instead of 80 bytes through the L2 -> CU path that bounds this kernel, and two weight
    // tables for the alternating parity of the rows).  The five chunks of a sample are five consecutive 16-byte LDS slots:
    // stride 80 B = 20 banks, which takes the 16 lanes of a ds_read_b128 group through all 64 banks.
    // does any stencil of this tile end within a chunk of the end of the table stack?  (block-uniform; almost never)
    const long my_last = my_seg0 < 0 ? 0 : (my_step > 0 ? my_seg0 + 9L * my_step : my_seg0) + 10;
    const int near_end = __syncthreads_or(my_last >= tab_elems);
    // the five (sample, chunk) pairs this thread fetches for every stencil row.  A wave stages the chunks of its
    // own 64 samples (work item w = lane + 64 q of the wave's 320), so that producer and consumer of an LDS row are
    // the same wave: no workgroup barrier in the row loop, only counted vmcnt waits, and the four waves drift freely.
    const double *at[NCH];  // the chunk of the next row
    int st1[NCH];           // and the element step between rows
#pragma unroll
    for (int q = 0; q < NCH; q++) {
        const int w = (tid & 63) + 64 * q, sm = (tid & ~63) + w / NCH, ch = w % NCH;
        const long e0 = seg0[sm];
        const int stp = rstep[sm];
        const bool on = e0 >= 0;  // samples without a stencil fetch element 0: harmless, and the fast path stays branch-free
        at[q] = tables + (on ? e0 + 2 * ch : 0);
        st1[q] = on ? stp : 0;
    }
    const int wave_base = (tid & ~63) * SEG_W;  // this wave's 64 x 10 doubles of a row buffer
    auto stage = [&](double *buf) {  // branch-free: every lane issues its five chunk loads
#pragma unroll
        for (int q = 0; q < NCH; q++) {
            IMCOM_GLDS16(at[q], buf + 128 * q + wave_base);  // wave-uniform LDS base; lane l lands at +2 l
            at[q] += st1[q];
        }
    };
    auto stage_checked = [&](double *buf) {  // tiles whose stencils reach the end of the table stack
#pragma unroll
        for (int q = 0; q < NCH; q++) {
            double *dst = buf + 128 * q + wave_base;
            const long a_ = at[q] - tables;
            if (a_ >= 0 && a_ + 1 < tab_elems) IMCOM_GLDS16(at[q], dst);
            else { dst[2 * (tid & 63)] = (a_ >= 0 && a_ < tab_elems) ? tables[a_] : 0.0; dst[2 * (tid & 63) + 1] = 0.0; }
            at[q] += st1[q];
        }
    };
    auto consume = [&](const double *buf, double wyr) {
        const f64x2 *row = (const f64x2 *)(buf + tid * SEG_W);
        f64x2 c[NCH];
#pragma unroll
        for (int q = 0; q < NCH; q++) c[q] = row[q];
        double strip = 0.0;
#pragma unroll
        for (int m = 0; m < 10; m++) strip += wt[m] * c[m >> 1][m & 1];
        val += strip * wyr;
    };
    if (!near_end) {
        stage(seg[0]);
        if (A_RING > 2) stage(seg[1]);
    }
    // the interpolation weights, formed while the first stencil row is in flight
    if (active) {
        d5512_getw(wx, fx);
        d5512_getw(wy, fy);
        if (rev) {  // reversed taps: flip the x weights once; the sum below then runs k = 9..0,
                    // which is the reference's order over the flipped table's columns
#pragma unroll
            for (int c = 0; c < 5; c++) { const double t = wx[c]; wx[c] = wx[9 - c]; wx[9 - c] = t; }
        }
    }
#pragma unroll
    for (int m = 0; m < 10; m++) wt[m] = active ? wx[m] : 0.0;
    if (!near_end) {
        // One row (five DMA instructions of this wave) is in flight while the previous one is consumed: vmcnt(5)
        // = the older row has landed.  lgkmcnt(0) before a buffer is refilled: its LDS reads have returned.
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const int ahead = r + A_RING - 1;  // the row issued now; rows r+1 .. ahead are in flight while r is consumed
            if (ahead < 10) stage(seg[ahead % A_RING]);
            const int inflight = (ahead < 10 ? ahead : 9) - r;
            if (inflight == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (inflight == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            consume(seg[r % A_RING], wy[r]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
        stage_checked(seg[0]);
        __syncthreads();
#pragma unroll 1
        for (int rp = 0; rp < 5; rp++) {
            stage_checked(seg[1]);
            double wa = 0.0, wb = 0.0;
#pragma unroll
            for (int q = 0; q < 5; q++) { wa = rp == q ? wy[2 * q] : wa; wb = rp == q ? wy[2 * q + 1] : wb; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            consume(seg[0], wa);
            __syncthreads();
            if (rp < 4) stage_checked(seg[0]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            consume(seg[1], wb);
            __syncthreads();
        }
    }
    val += pen;
    if (i >= ns || j >= ns) val = (i == j) ? 1.0 : 0.0;  // identity padding
    tile[li][lj] = val;
    __syncthreads();
    if (ti == tj) {
        const double v = (lj >= li) ? tile[li][lj] : tile[lj][li];
        if (i < ldn && j < ldn) As[(long)i * ldn + j] = v;
    } else {
        if (i < ldn && j < ldn) As[(long)i * ldn + j] = val;
        // mirrored tile, written row-wise: element (tj*16 + li, ti*16 + lj) = tile[lj][li]
        const int mi = tj * 16 + li, mj = ti * 16 + lj;
        if (mi < ldn && mj < ldn) As[(long)mi * ldn + mj] = tile[lj][li];
    }
}

// ---- per-stamp tile order by PSF pair (counting sort: key, exclusive scan per stamp, scatter) -------------------------------
__global__ void a_tile_key_kernel(const int *__restrict__ n, int ldn, const int *__restrict__ psf, int npsf_max, int ntile, long total,
                                  int *__restrict__ key, int *__restrict__ hist)
{
    const long g = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (g >= total) return;
    const long ntri = (long)ntile * (ntile + 1) / 2;
    const int s = (int)(g / ntri), nbin = npsf_max * npsf_max + 1;
    int ti, tj;
    tile_index(g - (long)s * ntri, ntile, ti, tj);
    const int ns = n[s];
    int k = nbin - 1;  // tiles beyond the stamp's pixels
    if (ti * 16 < ns && tj * 16 < ns) k = psf[(long)s * ldn + ti * 16] * npsf_max + psf[(long)s * ldn + tj * 16];
    key[g] = k;
    atomicAdd(hist + (long)s * nbin + k, 1);
}

__global__ __launch_bounds__(256) void a_tile_scan_kernel(int *__restrict__ hist, int nbin)  // one workgroup per stamp: exclusive scan of its bins in place
{
    __shared__ int part[256];
    int *h = hist + (long)blockIdx.x * nbin;
    const int per = (nbin + 255) / 256, b0 = threadIdx.x * per;
    int sum = 0;
    for (int q = 0; q < per; q++) sum += b0 + q < nbin ? h[b0 + q] : 0;
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int q = 0; q < per; q++)
        if (b0 + q < nbin) { const int v = h[b0 + q]; h[b0 + q] = run; run += v; }
}

__global__ void a_tile_scatter_kernel(const int *__restrict__ key, int *__restrict__ cursor, int npsf_max, int ntile, long total,
                                      int *__restrict__ desc)
{
    const long g = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (g >= total) return;
    const long ntri = (long)ntile * (ntile + 1) / 2;
    const int s = (int)(g / ntri), nbin = npsf_max * npsf_max + 1;
    desc[(long)s * ntri + atomicAdd(cursor + (long)s * nbin + key[g], 1)] = (int)(g - (long)s * ntri);  // the order inside a bucket decides only the schedule
}

#ifdef IMCOM_DEV
int launch_build_A_win(imcom_ctx *ctx, int batch, const int *n_dev, int ldn, const double *x, const double *y, const int *psf,
                       const double *tables, int ntab, int ng, double nc, double dscale, const int *pair_tab, const double *pair_pen,
                       int npsf_max, double *A);  // build_a_win.hip
#endif

int launch_build_A(imcom_ctx *ctx, int batch, const int *n_dev, int ldn, const double *x, const double *y,
                   const int *psf, const double *tables, int ntab, int ng, double nc, double dscale,
                   const int *pair_tab, const double *pair_pen, int npsf_max, double *A)
{
#ifdef IMCOM_DEV
    // developer build (make DEV=1): IMCOM_BUILD_A=window selects the experimental LDS-window builder (build_a_win.hip:
    // parity-tested, but slower than this file's per-sample DMA builder on rotated exposures -- DESIGN.md, "A builder, round 2")
    static const bool window = getenv("IMCOM_BUILD_A") && !strcmp(getenv("IMCOM_BUILD_A"), "window");
    if (window) return launch_build_A_win(ctx, batch, n_dev, ldn, x, y, psf, tables, ntab, ng, nc, dscale, pair_tab, pair_pen, npsf_max, A);
#endif
    IMCOM_REQUIRE(ntab >= 1 && ntab <= PAIR_MASK + 1, "table stack of %d tables (pair codes carry 28-bit table indices)", ntab);
    const int nt = (ldn + 15) / 16;
    const long ntri = (long)nt * (nt + 1) / 2;
    const long ngrid = (ntri + 7) / 8 * 8;  // padded to a multiple of 8 for the XCD-aware tile order
    int *desc = nullptr;
    // Every stamp's tiles ordered by the PSF pair of their first sample (three small kernels: counting sort per stamp), so that the
    // tiles an XCD works on at one time read ONE table: 27.85 against 29.25 ms per 256 cfg-2 stamps, three A/B pairs on one box
    // (ordering the whole BATCH by table instead lost 1.8x: profiles/r03_negative_results.txt).  IMCOM_A_ORDER=rows: row by row.
    static const bool by_pair = !(getenv("IMCOM_A_ORDER") && !strcmp(getenv("IMCOM_A_ORDER"), "rows"));
    if (by_pair) {
        const long total = ntri * batch;
        const int nbin = npsf_max * npsf_max + 1;
        int *key = (int *)ws_take(ctx, (size_t)total * 4), *hist = (int *)ws_take(ctx, (size_t)batch * nbin * 4);
        desc = (int *)ws_take(ctx, (size_t)total * 4);
        if (key && hist && desc) {
            IMCOM_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)batch * nbin * 4, ctx->stream));
            hipLaunchKernelGGL(a_tile_key_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, n_dev, ldn, psf, npsf_max, nt, total, key, hist);
            hipLaunchKernelGGL(a_tile_scan_kernel, dim3(batch), dim3(256), 0, ctx->stream, hist, nbin);
            hipLaunchKernelGGL(a_tile_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, key, hist, npsf_max, nt, total, desc);
            IMCOM_TRY(check_launch("build_A tile order"));
        } else desc = nullptr;
    }
    hipLaunchKernelGGL(build_A_kernel, dim3((unsigned)ngrid, batch), dim3(256), 0, ctx->stream, n_dev, ldn, x, y, psf,
                       tables, (long)ntab * ng * ng, ng, nc, dscale, pair_tab, pair_pen, npsf_max, A, nt, (const int *)desc);
    return check_launch("build_A_kernel");
}

}  // namespace imcom
