// interp.hip -- D5512 (10x10-tap) table interpolation kernels and the A / B builders.
//
// Replaces furry_parakeet.pyimcom_croutines.{iD5512C, iD5512C_sym, gridD5512C} (in-tree spec:
// reference src/pyimcom/routine.py:29-338) and, for the device-resident stamp path,
// PSFOvl._call_ii_self/_call_ii_cross/_call_io_cross (psfutil.py:1401-1732) + the A/B assembly of
// OutStamp._build_system_matrices (coadd.py:1027-1082).
//
// These kernels are gather + fp64 FMA work: no MFMA.  The B builder is separable (every input pixel
// sees a regular n2f x n2f grid of output pixels): the x-pass result for the window of table rows the
// stamp touches is kept in LDS and the y-pass reads it back, 5x fewer table reads than the direct
// form, in the reference's own summation order (inner sum over x taps, outer over y taps).
#include "common.h"
#include "d5512.h"

namespace imcom {

typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));  // two doubles at any double boundary

// ------------------------------------------------------------------------------------------------
__global__ void d5512_getw_kernel(const double *__restrict__ fh, long n, double *__restrict__ w)
{
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double ww[10];
    d5512_getw(ww, fh[i]);
#pragma unroll
    for (int k = 0; k < 10; k++) w[i * 10 + k] = ww[k];
}

// routine.py:125-181
__global__ void interp_d5512_kernel(const double *__restrict__ infunc, int nlayer, int ngy, int ngx,
                                    const double *__restrict__ xpos, const double *__restrict__ ypos,
                                    long nout, double *__restrict__ fhatout)
{
    long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (p >= nout) return;
    const double x = xpos[p], y = ypos[p];
    const int xi = to_cell(x), yi = to_cell(y);
    if (xi < 4 || xi >= ngx - 5 || yi < 4 || yi >= ngy - 5) return;  // output untouched
    double wx[10], wy[10];
    d5512_getw(wx, x - xi - 0.5);
    d5512_getw(wy, y - yi - 0.5);
    for (int l = 0; l < nlayer; l++)
        fhatout[(long)l * nout + p] =
            stencil(infunc + ((long)l * ngy + (yi - 4)) * ngx + (xi - 4), ngx, 1, wx, wy);
}

// routine.py:184-253: upper triangle (a <= b) interpolated, lower = copy of upper (even when the
// upper point was off-grid and kept its previous value, lines 249-253)
__global__ void interp_d5512_sym_kernel(const double *__restrict__ infunc, int nlayer, int ngy, int ngx,
                                        const double *__restrict__ xpos,
                                        const double *__restrict__ ypos, long nout, long sq,
                                        double *__restrict__ fhatout)
{
    long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (t >= sq * sq) return;
    const long a = t / sq, b = t % sq;
    if (b < a) return;
    const long p = a * sq + b, pm = b * sq + a;
    const double x = xpos[p], y = ypos[p];
    const int xi = to_cell(x), yi = to_cell(y);
    const bool on = !(xi < 4 || xi >= ngx - 5 || yi < 4 || yi >= ngy - 5);
    double wx[10], wy[10];
    if (on) {
        d5512_getw(wx, x - xi - 0.5);
        d5512_getw(wy, y - yi - 0.5);
    }
    for (int l = 0; l < nlayer; l++) {
        double v;
        if (on) {
            v = stencil(infunc + ((long)l * ngy + (yi - 4)) * ngx + (xi - 4), ngx, 1, wx, wy);
            fhatout[(long)l * nout + p] = v;
        } else
            v = fhatout[(long)l * nout + p];
        if (b != a) fhatout[(long)l * nout + pm] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Separable grid interpolation (routine.py:256-338) for one input pixel per workgroup.
//   xs[nxo], ys[nyo] : table coordinates of the output columns / rows for this pixel
// LDS: wx[nxo][10], wy[nyo][10], xi[nxo], yi[nyo], tmp[nrows][nxo] (the x-pass of the touched rows).
// Falls back to the direct 100-tap form when the touched row window does not fit `max_rows`.
template <typename XF, typename YF>
__device__ __forceinline__ void grid_pixel(const double *__restrict__ f, int ngy, int ngx, int nxo, int nyo,
                                           XF xcoord, YF ycoord, double *__restrict__ out, double *sm_d,
                                           int *sm_i, int max_rows)
{
    const int tid = threadIdx.x, nth = blockDim.x;
    double *wxs = sm_d, *wys = sm_d + 10 * nxo, *tmp = wys + 10 * nyo;
    int *xis = sm_i, *yis = sm_i + nxo, *red = yis + nyo;  // red[0]=rlo, red[1]=rhi
    if (tid == 0) { red[0] = 0x7fffffff; red[1] = -0x7fffffff; }
    __syncthreads();
    int lo = 0x7fffffff, hi = -0x7fffffff;  // touched table rows seen by this thread
    for (int t = tid; t < nxo + nyo; t += nth) {
        const bool isx = t < nxo;
        const int idx = isx ? t : t - nxo;
        const double v = isx ? xcoord(idx) : ycoord(idx);
        int c = to_cell(v);
        const int lim = isx ? ngx : ngy;
        double w[10];
        if (c < 4 || c >= lim - 5) {  // off the grid: zero weights, cell 4 (routine.py:306-323)
            c = 4;
#pragma unroll
            for (int k = 0; k < 10; k++) w[k] = 0.0;
            if (!isx) c = -1;  // marker: row window is taken from valid rows only
        } else {
            d5512_getw(w, v - c - 0.5);
            if (!isx) { lo = min(lo, c - 4); hi = max(hi, c + 5); }
        }
        double *dst = isx ? wxs + 10 * idx : wys + 10 * idx;
#pragma unroll
        for (int k = 0; k < 10; k++) dst[k] = w[k];
        (isx ? xis : yis)[idx] = c;
    }
    // one LDS atomic pair per wave (48 lanes hitting the same two words one after the other cost a quarter of this kernel)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, 64));
        hi = max(hi, __shfl_xor(hi, off, 64));
    }
    if ((tid & 63) == 0 && hi >= lo) { atomicMin(&red[0], lo); atomicMax(&red[1], hi); }
    __syncthreads();
    const int rlo = red[0], rhi = red[1];
    const int nrows = rhi - rlo + 1;
    if (rhi < rlo) {  // no valid row at all: every output is exactly zero
        for (int t = tid; t < nxo * nyo; t += nth) out[t] = 0.0;
        return;
    }
    if (nrows <= max_rows) {
        // x-pass over the touched table rows.  A thread keeps one output column -- its ten weights in registers, its tap address
        // advancing by whole rows -- and walks the rows three at a time (fifteen 16-byte loads in flight: the pass waits on
        // their latency; one row at a time left the five loads of an iteration alone in flight).
        // (ten taps as five 16-byte loads at an 8-byte aligned address: global memory takes them)
        if (nxo <= nth) {
            const int nlane = nth / nxo, ix = tid % nxo, rl = tid / nxo;
            if (rl < nlane) {
                double w[10];
#pragma unroll
                for (int j = 0; j < 10; j++) w[j] = wxs[10 * ix + j];
                const double *col = f + (long)rlo * ngx + (xis[ix] - 4);
                constexpr int U = 3;  // rows in flight per thread
                int r = rl;
                for (; r + (U - 1) * nlane < nrows; r += U * nlane) {
                    f64x2u a[U][5];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const f64x2u *ra = (const f64x2u *)(col + (long)(r + u * nlane) * ngx);
#pragma unroll
                        for (int q = 0; q < 5; q++) a[u][q] = ra[q];
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        double sa = 0.0;
#pragma unroll
                        for (int j = 0; j < 10; j++) sa += w[j] * a[u][j >> 1][j & 1];
                        tmp[(r + u * nlane) * nxo + ix] = sa;
                    }
                }
                for (; r < nrows; r += nlane) {
                    const f64x2u *ra = (const f64x2u *)(col + (long)r * ngx);
                    f64x2u a[5];
#pragma unroll
                    for (int q = 0; q < 5; q++) a[q] = ra[q];
                    double sa = 0.0;
#pragma unroll
                    for (int j = 0; j < 10; j++) sa += w[j] * a[j >> 1][j & 1];
                    tmp[r * nxo + ix] = sa;
                }
            }
        } else
        for (int t = tid; t < nrows * nxo; t += nth) {
            const int r = t / nxo, ix = t - r * nxo;
            const f64x2u *row = (const f64x2u *)(f + (long)(rlo + r) * ngx + (xis[ix] - 4));
            const double *w = wxs + 10 * ix;
            f64x2u c[5];
#pragma unroll
            for (int q = 0; q < 5; q++) c[q] = row[q];
            double strip = 0.0;
#pragma unroll
            for (int j = 0; j < 10; j++) strip += w[j] * c[j >> 1][j & 1];
            tmp[t] = strip;
        }
        __syncthreads();
        if (nxo <= nth) {
            const int nlane = nth / nxo, ix = tid % nxo, yl = tid / nxo;
            if (yl < nlane)
                for (int iy = yl; iy < nyo; iy += nlane) {
                    const int yc = yis[iy];
                    const double *w = wys + 10 * iy;
                    const double *col = tmp + (yc < 0 ? 0 : yc - 4 - rlo) * nxo + ix;  // invalid rows carry zero weights: any staged rows do
                    double o = 0.0;
#pragma unroll
                    for (int i = 0; i < 10; i++) o += col[i * nxo] * w[i];
                    out[iy * nxo + ix] = o;
                }
        } else
        for (int t = tid; t < nxo * nyo; t += nth) {
            const int iy = t / nxo, ix = t - iy * nxo;
            const int yc = yis[iy];
            const double *w = wys + 10 * iy;
            const int r0 = (yc < 0 ? 0 : yc - 4 - rlo);  // invalid rows carry zero weights: any staged rows do
            double o = 0.0;
#pragma unroll
            for (int i = 0; i < 10; i++) o += tmp[(r0 + i) * nxo + ix] * w[i];
            out[t] = o;
        }
    } else {
        for (int t = tid; t < nxo * nyo; t += nth) {
            const int iy = t / nxo, ix = t - iy * nxo;
            const int yc = yis[iy] < 0 ? 4 : yis[iy];
            const double *w = wys + 10 * iy, *wxp = wxs + 10 * ix;
            double o = 0.0;
            for (int i = 0; i < 10; i++) {
                const double *row = f + (long)(yc - 4 + i) * ngx + (xis[ix] - 4);
                double strip = 0.0;
#pragma unroll
                for (int j = 0; j < 10; j++) strip += wxp[j] * row[j];
                o += strip * w[i];
            }
            out[t] = o;
        }
    }
}

__global__ __launch_bounds__(256) void grid_d5512_kernel(const double *__restrict__ infunc, int ngy, int ngx,
                                                         const double *__restrict__ xpos,
                                                         const double *__restrict__ ypos, long npi, int nxo,
                                                         int nyo, double *__restrict__ fhatout, int max_rows)
{
    extern __shared__ double smem[];
    const long p = blockIdx.x;
    if (p >= npi) return;
    double *sm_d = smem;
    int *sm_i = (int *)(smem + 10 * (nxo + nyo) + (size_t)max_rows * nxo);
    const double *xp = xpos + p * nxo, *yp = ypos + p * nyo;
    grid_pixel(infunc, ngy, ngx, nxo, nyo, [&](int i) { return xp[i]; }, [&](int i) { return yp[i]; },
               fhatout + p * (long)nxo * nyo, sm_d, sm_i, max_rows);
}

// ------------------------------------------------------------------------------------------------
// B builder (psfutil.py:1497-1595): one workgroup per (input pixel, stamp).
//   ddx = (x_in - x_out[ix]) / dscale + nc (+6 for the zero border), ddy likewise (1536-1541, 1581-1582)
__global__ __launch_bounds__(256) void build_B_kernel(const int *__restrict__ n, int ldn,
                                                      const double *__restrict__ x,
                                                      const double *__restrict__ y,
                                                      const int *__restrict__ psf,
                                                      const double *__restrict__ tables, int ng,
                                                      double nc, double dscale,
                                                      const int *__restrict__ io_tab, int npsf_max,
                                                      const double *__restrict__ out_x0,
                                                      const double *__restrict__ out_y0, int n2f, int ldm,
                                                      double *__restrict__ Bt, int max_rows)
{
    extern __shared__ double smem[];
    const int s = blockIdx.y, i = blockIdx.x;
    double *dst = Bt + ((long)s * ldn + i) * ldm;
    const int m = n2f * n2f;
    if (i >= n[s]) {  // padding rows are zero
        for (int t = threadIdx.x; t < ldm; t += blockDim.x) dst[t] = 0.0;
        return;
    }
    const double xin = x[(long)s * ldn + i], yin = y[(long)s * ldn + i];
    const int tab = io_tab[(long)s * npsf_max + psf[(long)s * ldn + i]];
    const double *f = tables + (long)tab * ng * ng;
    const double x0 = out_x0[s], y0 = out_y0[s];
    double *sm_d = smem;
    int *sm_i = (int *)(smem + 20 * n2f + (size_t)max_rows * n2f);
    grid_pixel(f, ng, ng, n2f, n2f,
               [&](int ix) { double d = xin - (x0 + ix); d /= dscale; d += nc; return d + 6.0; },
               [&](int iy) { double d = yin - (y0 + iy); d /= dscale; d += nc; return d + 6.0; },
               dst, sm_d, sm_i, max_rows);
    for (int t = m + threadIdx.x; t < ldm; t += blockDim.x) dst[t] = 0.0;
}

// ------------------------------------------------------------------------------------------------
int launch_getw(imcom_ctx *ctx, const double *fh, long n, double *w)
{
    if (n <= 0) return IMCOM_OK;
    hipLaunchKernelGGL(d5512_getw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, fh, n, w);
    return check_launch("d5512_getw_kernel");
}

int launch_interp(imcom_ctx *ctx, const double *infunc, int nlayer, int ngy, int ngx, const double *xpos,
                  const double *ypos, long nout, double *fhatout, int sym)
{
    if (nout <= 0) return IMCOM_OK;
    if (!sym) {
        hipLaunchKernelGGL(interp_d5512_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, ctx->stream,
                           infunc, nlayer, ngy, ngx, xpos, ypos, nout, fhatout);
        return check_launch("interp_d5512_kernel");
    }
    long sq = (long)sqrt((double)(nout + 1));  // routine.py:214
    hipLaunchKernelGGL(interp_d5512_sym_kernel, dim3((unsigned)((sq * sq + 255) / 256)), dim3(256), 0,
                       ctx->stream, infunc, nlayer, ngy, ngx, xpos, ypos, nout, sq, fhatout);
    return check_launch("interp_d5512_sym_kernel");
}

// rows_hint > 0: the caller knows how many table rows one pixel's output grid can touch (regular grids);
// reserving just that keeps several workgroups resident per CU.  Otherwise a 96 KB budget is used.
static int grid_lds(int nxo, int nyo, int rows_hint, int *max_rows, size_t *bytes)
{
    const size_t fixed = (size_t)(10 * (nxo + nyo)) * 8 + (size_t)(nxo + nyo + 4) * 4;
    const size_t budget = 96 * 1024;
    long rows = fixed < budget ? (long)((budget - fixed) / ((size_t)nxo * 8)) : 0;
    if (rows_hint > 0 && rows_hint < rows) rows = rows_hint;
    if (rows > 4096) rows = 4096;
    if (rows < 10) rows = 0;  // too wide for the LDS form: direct evaluation
    *max_rows = (int)rows;
    *bytes = fixed + (size_t)rows * nxo * 8 + 16;
    return IMCOM_OK;
}

int launch_grid(imcom_ctx *ctx, const double *infunc, int ngy, int ngx, const double *xpos, const double *ypos,
                long npi, int nxo, int nyo, double *fhatout)
{
    if (npi <= 0) return IMCOM_OK;
    int max_rows; size_t bytes;
    grid_lds(nxo, nyo, 0, &max_rows, &bytes);
    IMCOM_REQUIRE(bytes <= 160 * 1024, "gridD5512C: output grid %d x %d too large for the LDS form", nxo, nyo);
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)grid_d5512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(grid_d5512_kernel, dim3((unsigned)npi), dim3(256), bytes, ctx->stream, infunc, ngy, ngx,
                       xpos, ypos, npi, nxo, nyo, fhatout, max_rows);
    return check_launch("grid_d5512_kernel");
}

int launch_build_B(imcom_ctx *ctx, int batch, const int *n_dev, int ldn, const double *x, const double *y,
                   const int *psf, const double *tables, int ng, double nc, double dscale, const int *io_tab,
                   int npsf_max, const double *out_x0, const double *out_y0, int n2f, int ldm, double *Bt)
{
    int max_rows; size_t bytes;
    // output pixels are 1/dscale table samples apart: (n2f-1)/dscale + 10 taps (+2 for rounding at both ends)
    grid_lds(n2f, n2f, (int)((n2f - 1) / dscale) + 13, &max_rows, &bytes);
    IMCOM_REQUIRE(bytes <= 160 * 1024, "build_B: n2f=%d too large", n2f);
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)build_B_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(build_B_kernel, dim3(ldn, batch), dim3(256), bytes, ctx->stream, n_dev, ldn, x, y, psf,
                       tables, ng, nc, dscale, io_tab, npsf_max, out_x0, out_y0, n2f, ldm, Bt, max_rows);
    return check_launch("build_B_kernel");
}


}  // namespace imcom
