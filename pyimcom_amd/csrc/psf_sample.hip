// psf_sample.hip -- PSF images onto the output-aligned sample grid, and the analytic target PSFs.
//
// Reference (src/pyimcom/psfutil.py):
//   PSFGrp._sample_psf 709-795           pad by 6, interpolate (iD5512C at given positions / gridD5512C on the unrotated grid)
//   PSFGrp.__init__ 650-656              circular cut-out (ro < nc + 0.5) and normalisation to unit sum
//   OutPSF.psf_gaussian 117-146          Gaussian spot
//   OutPSF.psf_simple_airy 148-223       (obscured) Airy spot, optionally convolved with a top-hat and a Gaussian
// The convolution of the Airy spot is a multiplication by a real, even, separable filter in Fourier space, i.e.
// circulant matrices applied from both sides: out = C I C^T with C[i][j] = k[(i - j) mod npad],
// k[d] = (1/npad) sum_u h(u) cos(2 pi u d / npad) -- two products on the fp64 MFMA GEMM, exact twiddles.
#include "common.h"
#include "d5512.h"
#include "launchers.h"

namespace imcom {

// padded[p][r][c] = psf[p][r-6][c-6] (zero border), ng = n + 12
__global__ void pad6_kernel(const double *__restrict__ psf, int ny, int nx, double *__restrict__ out)
{
    const int p = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    const int gy = ny + 12, gx = nx + 12;
    if (c >= gx) return;
    const int rr = r - 6, cc = c - 6;
    out[((long)p * gy + r) * gx + c] = (rr >= 0 && rr < ny && cc >= 0 && cc < nx) ? psf[((long)p * ny + rr) * nx + cc] : 0.0;
}

// All PSFs of a call in one launch: PSF p = blockIdx.y is interpolated at its own positions (psfutil.py:777-783,
// routine.py:125-181 with one layer); samples whose stencil leaves the padded image keep the zero they were set to.
__global__ void sample_rot_kernel(const double *__restrict__ pad, int gy, int gx, const double *__restrict__ yxco, long npts, double xctr,
                                  double yctr, double *__restrict__ out)
{
    const int p = blockIdx.y;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const double *co = yxco + (long)p * 2 * npts;
    const double y = co[i] + yctr + 6.0, x = co[npts + i] + xctr + 6.0;
    const int xi = to_cell(x), yi = to_cell(y);
    if (xi < 4 || xi >= gx - 5 || yi < 4 || yi >= gy - 5) return;
    double wx[10], wy[10];
    d5512_getw(wx, x - xi - 0.5);
    d5512_getw(wy, y - yi - 0.5);
    out[(long)p * npts + i] = stencil(pad + ((long)p * gy + (yi - 4)) * gx + (xi - 4), gx, 1, wx, wy);
}

// unrotated grid: lin[i] = i - (nsamp-1)/2 (PSFGrp.yxo), positions lin + ctr + 6   (psfutil.py:788-790)
__global__ void grid_pos_kernel(int nsamp, double xctr, double yctr, double *__restrict__ xpos, double *__restrict__ ypos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsamp) return;
    const double lin = (double)i - 0.5 * (double)(nsamp - 1);
    xpos[i] = lin + xctr + 6.0;
    ypos[i] = lin + yctr + 6.0;
}

// circular cut-out and per-PSF sum, then scale.  One block per (row, PSF) leaves the row's sum (rows = fixed units of the
// reduction: the result does not depend on how many blocks run at once); a second pass adds the rows of a PSF in order.
__global__ __launch_bounds__(256) void psf_circ_rowsum_kernel(double *__restrict__ arr, int nsamp, int circ, double *__restrict__ rowsum)
{
    __shared__ double red[4];
    const int r = blockIdx.x, p = blockIdx.y;
    const double c0 = 0.5 * (double)(nsamp - 1), rmax = (double)(nsamp / 2) + 0.5;
    double *row = arr + ((long)p * nsamp + r) * nsamp;
    double acc = 0.0;
    for (int c = threadIdx.x; c < nsamp; c += 256) {
        double v = row[c];
        if (circ && !(hypot((double)r - c0, (double)c - c0) < rmax)) {
            v = 0.0;
            row[c] = 0.0;
        }
        acc += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) rowsum[(long)p * nsamp + r] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void psf_sum_rows_kernel(const double *__restrict__ rowsum, int nsamp, double *__restrict__ sums)
{
    __shared__ double red[4];
    const int p = blockIdx.x;
    double acc = 0.0;
    for (int r = threadIdx.x; r < nsamp; r += 256) acc += rowsum[(long)p * nsamp + r];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[p] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void psf_scale_kernel(double *__restrict__ arr, long per, const double *__restrict__ sums)
{
    const int p = blockIdx.y;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < per) arr[p * per + i] /= sums[p];
}

__global__ void gaussian_kernel(int n, double sigmax, double sigmay, double *__restrict__ out)
{
    const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    // np.mgrid[(1-n)/2/s : (n-1)/2/s : n*1j]: start + i * step with step = (stop - start) / (n - 1)
    const double y0 = 0.5 * (1 - n) / sigmay, y1 = 0.5 * (n - 1) / sigmay, x0 = 0.5 * (1 - n) / sigmax, x1 = 0.5 * (n - 1) / sigmax;
    const double y = n > 1 ? y0 + r * ((y1 - y0) / (n - 1)) : y0, x = n > 1 ? x0 + c * ((x1 - x0) / (n - 1)) : x0;
    out[(long)r * n + c] = exp(-0.5 * (x * x + y * y)) / (2.0 * M_PI * sigmax * sigmay);
}

// Airy intensity on the npad x npad grid (psfutil.py:189-203), written into the padded GEMM operand [P][P]
__global__ void airy_kernel(int npad, int P, double ldp, double obsc, double *__restrict__ I)
{
    const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P) return;
    double v = 0.0;
    if (r < npad && c < npad) {
        const double y = (double)r - 0.5 * (double)(npad - 1), x = (double)c - 0.5 * (double)(npad - 1);
        const double pr = M_PI * (sqrt(x * x + y * y) / ldp);
        double a = j0(pr) + jn(2, pr);
        if (obsc != 0.0) a -= obsc * obsc * (j0(pr * obsc) + jn(2, pr * obsc));
        v = a * a / (4.0 * ldp * ldp * (1.0 - obsc * obsc)) * M_PI;
    }
    I[(long)r * P + c] = v;
}

// circulant of the separable filter h(u) = exp(-2 pi^2 u^2 sigma^2) sinc(u tophat), u_k = k/npad (minus 1 in the upper half)
__global__ __launch_bounds__(256) void airy_filter_kernel(int npad, double sigma, double tophat, double *__restrict__ kvec)
{
    __shared__ double red[4];
    const int d = blockIdx.x;
    double acc = 0.0;
    for (int k = threadIdx.x; k < npad; k += 256) {
        double u = (double)k / (double)npad;
        if (k >= npad - npad / 2) u -= 1.0;
        const double us = u * sigma, ut = u * tophat;
        const double snc = ut == 0.0 ? 1.0 : sinpi(ut) / (M_PI * ut);
        const double h = exp(-2.0 * M_PI * M_PI * (us * us)) * snc;
        const long kd = ((long)k * d) % npad;
        acc += h * cospi(2.0 * (double)kd / (double)npad);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) kvec[d] = ((red[0] + red[1]) + (red[2] + red[3])) / (double)npad;
}

__global__ void circulant_kernel(const double *__restrict__ kvec, int npad, int P, double *__restrict__ Cm)
{
    const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P) return;
    double v = 0.0;
    if (r < npad && c < npad) v = kvec[(r - c + npad) % npad];
    Cm[(long)r * P + c] = v;
}

__global__ void crop_kernel(const double *__restrict__ Z, int P, int kp, int n, double *__restrict__ out)
{
    const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    out[(long)r * n + c] = Z[(long)(r + kp) * P + (c + kp)];
}

// smooth_and_pad operand: I[p][r][c] = in[p][r - npad][c - npad] inside, zero elsewhere (incl. the GEMM padding)
__global__ void pad_rect_kernel(const double *__restrict__ in, int ny, int nx, int npad, int Py, int Px, double *__restrict__ I)
{
    const int p = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Px) return;
    const int rr = r - npad, cc = c - npad;
    I[((long)p * Py + r) * Px + c] = (rr >= 0 && rr < ny && cc >= 0 && cc < nx) ? in[((long)p * ny + rr) * nx + cc] : 0.0;
}

__global__ void crop_rect_kernel(const double *__restrict__ Z, int Py, int Px, int nyy, int nxx, double *__restrict__ out)
{
    const int p = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nxx) return;
    out[((long)p * nyy + r) * nxx + c] = Z[((long)p * Py + r) * Px + c];
}

// Sampling positions from a coarse lattice (round 6).  The reference evaluates the WCS chain outpix2world2inpix at all nsamp^2 =
// 146 689 sampling positions of a PSF group and exposure (psfutil.py:751-771); over the 5" the samples span that map is smooth to
// rounding at polynomial degree 16, so the host evaluates it on an L x L lattice of Chebyshev-Lobatto nodes and this kernel forms
// yxco[c][k][iy][ix] = sum_a sum_b W[iy][a] W[ix][b] lat[c][k][a][b], W[i][a] = the a-th Lagrange basis polynomial of the nodes at
// sample i (host, barycentric form).  One workgroup per (ROWS sample rows, coordinate plane): T[a][ix] = sum_b lat[a][b] W[ix][b] in
// LDS, then out[iy][ix] = sum_a W[iy][a] T[a][ix].
constexpr int LAT_ROWS = 32;
__global__ __launch_bounds__(256) void lattice_positions_kernel(const double *__restrict__ W, const double *__restrict__ lat, int L, int ns,
                                                                double *__restrict__ out)
{
    extern __shared__ double Tl[];  // [L][ns]
    const int plane = blockIdx.y, r0 = blockIdx.x * LAT_ROWS;
    const double *F = lat + (long)plane * L * L;
    for (int ix = threadIdx.x; ix < ns; ix += 256) {
        const double *w = W + (long)ix * L;
        for (int a = 0; a < L; a++) {
            double t = 0.0;
            for (int b = 0; b < L; b++) t += F[a * L + b] * w[b];
            Tl[a * ns + ix] = t;
        }
    }
    __syncthreads();
    double *o = out + (long)plane * ns * ns;
    for (int iy = r0; iy < min(r0 + LAT_ROWS, ns); iy++) {
        const double *w = W + (long)iy * L;
        for (int ix = threadIdx.x; ix < ns; ix += 256) {
            double v = 0.0;
            for (int a = 0; a < L; a++) v += w[a] * Tl[a * ns + ix];
            o[(long)iy * ns + ix] = v;
        }
    }
}

}  // namespace imcom

using namespace imcom;

static int ctx_ok3(imcom_ctx *ctx)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    return IMCOM_OK;
}

extern "C" int imcom_sample_psf(imcom_ctx *ctx, int n_psf, const double *psf, int ny, int nx, const double *yxco, int nsamp,
                                int psf_circ, int psf_norm, double *psf_arr, int memspace)
{
    IMCOM_TRY(ctx_ok3(ctx));
    IMCOM_REQUIRE(n_psf >= 1 && psf && psf_arr && ny >= 1 && nx >= 1 && nsamp >= 1, "bad arguments");
    const bool host = memspace == IMCOM_MEM_HOST;
    const long npts = (long)nsamp * nsamp, gy = ny + 12, gx = nx + 12;
    const size_t szin = (size_t)n_psf * ny * nx * 8, szout = (size_t)n_psf * npts * 8, szco = yxco ? (size_t)n_psf * 2 * npts * 8 : 0;
    size_t total = 65536 + (size_t)n_psf * gy * gx * 8 + 2 * (size_t)npts * 8 + (size_t)n_psf * 8 + (size_t)n_psf * nsamp * 8;
    if (host) total += szin + szout + szco + 1024;
    IMCOM_TRY(ws_reserve(ctx, total));
    const double *psf_d = psf, *co_d = yxco;
    double *out_d = psf_arr;
    if (host) {
        double *t = (double *)ws_take(ctx, szin);
        out_d = (double *)ws_take(ctx, szout);
        if (!t || !out_d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        IMCOM_HIP_CHECK(hipMemcpyAsync(t, psf, szin, hipMemcpyHostToDevice, ctx->stream));
        psf_d = t;
        if (yxco) {
            double *c = (double *)ws_take(ctx, szco);
            if (!c) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
            IMCOM_HIP_CHECK(hipMemcpyAsync(c, yxco, szco, hipMemcpyHostToDevice, ctx->stream));
            co_d = c;
        }
    }
    double *pad = (double *)ws_take(ctx, (size_t)n_psf * gy * gx * 8);
    double *xpos = (double *)ws_take(ctx, (size_t)npts * 8), *ypos = (double *)ws_take(ctx, (size_t)npts * 8);
    double *sums = (double *)ws_take(ctx, (size_t)n_psf * 8), *rowsum = (double *)ws_take(ctx, (size_t)n_psf * nsamp * 8);
    if (!pad || !xpos || !ypos || !sums || !rowsum) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    const double xctr = (nx - 1) / 2.0, yctr = (ny - 1) / 2.0;
    ProfScope ps(ctx, "psf_sample");
    hipLaunchKernelGGL(pad6_kernel, dim3((unsigned)((gx + 255) / 256), (unsigned)gy, n_psf), dim3(256), 0, ctx->stream, psf_d, ny, nx, pad);
    IMCOM_TRY(check_launch("pad6_kernel"));
    IMCOM_HIP_CHECK(hipMemsetAsync(out_d, 0, szout, ctx->stream));  // off-grid samples stay zero (psfutil.py:775, 785)
    if (!co_d) {
        hipLaunchKernelGGL(grid_pos_kernel, dim3((nsamp + 255) / 256), dim3(256), 0, ctx->stream, nsamp, xctr, yctr, xpos, ypos);
        IMCOM_TRY(check_launch("grid_pos_kernel"));
    }
    if (co_d) {
        hipLaunchKernelGGL(sample_rot_kernel, dim3((unsigned)((npts + 255) / 256), n_psf), dim3(256), 0, ctx->stream, (const double *)pad, (int)gy,
                           (int)gx, co_d, npts, xctr, yctr, out_d);
        IMCOM_TRY(check_launch("sample_rot_kernel"));
    } else {
        for (int p = 0; p < n_psf; p++)
            IMCOM_TRY(launch_grid(ctx, pad + (size_t)p * gy * gx, (int)gy, (int)gx, xpos, ypos, 1, nsamp, nsamp, out_d + (size_t)p * npts));
    }
    if (psf_circ || psf_norm) {
        hipLaunchKernelGGL(psf_circ_rowsum_kernel, dim3(nsamp, n_psf), dim3(256), 0, ctx->stream, out_d, nsamp, psf_circ ? 1 : 0, rowsum);
        hipLaunchKernelGGL(psf_sum_rows_kernel, dim3(n_psf), dim3(256), 0, ctx->stream, (const double *)rowsum, nsamp, sums);
        if (psf_norm)
            hipLaunchKernelGGL(psf_scale_kernel, dim3((unsigned)((npts + 255) / 256), n_psf), dim3(256), 0, ctx->stream, out_d, npts, sums);
        IMCOM_TRY(check_launch("psf circ/norm"));
    }
    if (host) {
        IMCOM_HIP_CHECK(hipMemcpyAsync(psf_arr, out_d, szout, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return IMCOM_OK;
}

extern "C" int imcom_lattice_positions(imcom_ctx *ctx, int count, int L, const double *W, const double *lattice, int nsamp, double *yxco,
                                       int memspace)
{
    IMCOM_TRY(ctx_ok3(ctx));
    IMCOM_REQUIRE(count >= 1 && L >= 2 && L <= 33 && W && lattice && yxco && nsamp >= 1 && nsamp <= 600, "bad arguments (2 <= L <= 33, nsamp <= 600)");
    const bool host = memspace == IMCOM_MEM_HOST;
    const size_t szW = (size_t)nsamp * L * 8, szL = (size_t)count * 2 * L * L * 8, szO = (size_t)count * 2 * nsamp * nsamp * 8;
    IMCOM_TRY(ws_reserve(ctx, szW + (host ? szL + szO : 0) + 8192));
    double *W_d = (double *)ws_take(ctx, szW);
    const double *lat_d = lattice;
    double *out_d = yxco;
    if (!W_d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    IMCOM_TRY(upload(ctx, W_d, W, (size_t)nsamp * L));
    if (host) {
        double *l = (double *)ws_take(ctx, szL);
        out_d = (double *)ws_take(ctx, szO);
        if (!l || !out_d) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        IMCOM_HIP_CHECK(hipMemcpyAsync(l, lattice, szL, hipMemcpyHostToDevice, ctx->stream));
        lat_d = l;
    }
    const size_t lds = (size_t)L * nsamp * 8;
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)lattice_positions_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ProfScope ps(ctx, "psf_sample");
        hipLaunchKernelGGL(lattice_positions_kernel, dim3((nsamp + LAT_ROWS - 1) / LAT_ROWS, 2 * count), dim3(256), lds, ctx->stream, (const double *)W_d, lat_d,
                           L, nsamp, out_d);
        IMCOM_TRY(check_launch("lattice_positions_kernel"));
    }
    if (host) {
        IMCOM_HIP_CHECK(hipMemcpyAsync(yxco, out_d, szO, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return IMCOM_OK;
}

extern "C" int imcom_psf_gaussian(imcom_ctx *ctx, int n, double sigmax, double sigmay, double *out, int memspace)
{
    IMCOM_TRY(ctx_ok3(ctx));
    IMCOM_REQUIRE(n >= 1 && out && sigmax > 0.0 && sigmay > 0.0, "bad arguments");
    const bool host = memspace == IMCOM_MEM_HOST;
    const size_t sz = (size_t)n * n * 8;
    IMCOM_TRY(ws_reserve(ctx, sz + 4096));
    double *o = host ? (double *)ws_take(ctx, sz) : out;
    if (!o) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    hipLaunchKernelGGL(gaussian_kernel, dim3((n + 255) / 256, n), dim3(256), 0, ctx->stream, n, sigmax, sigmay, o);
    IMCOM_TRY(check_launch("gaussian_kernel"));
    if (host) {
        IMCOM_HIP_CHECK(hipMemcpyAsync(out, o, sz, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return IMCOM_OK;
}

extern "C" int imcom_psf_simple_airy(imcom_ctx *ctx, int n, double ldp, double obsc, double tophat_conv, double sigma, double *out,
                                     int memspace)
{
    IMCOM_TRY(ctx_ok3(ctx));
    IMCOM_REQUIRE(n >= 1 && out && ldp > 0.0 && obsc >= 0.0 && obsc < 1.0 && tophat_conv >= 0.0 && sigma >= 0.0, "bad arguments");
    const bool host = memspace == IMCOM_MEM_HOST;
    const int kp = 1 + (int)ceil(tophat_conv + 6.0 * sigma), npad = n + 2 * kp;  // psfutil.py:185-186
    const int P = (int)align_up((size_t)npad, NB);
    const size_t szP = (size_t)P * P * 8, sz = (size_t)n * n * 8;
    IMCOM_TRY(ws_reserve(ctx, 4 * szP + (size_t)npad * 8 + sz + 8192));
    double *I = (double *)ws_take(ctx, szP), *Cm = (double *)ws_take(ctx, szP), *Y = (double *)ws_take(ctx, szP), *Z = (double *)ws_take(ctx, szP);
    double *kvec = (double *)ws_take(ctx, (size_t)npad * 8);
    double *o = host ? (double *)ws_take(ctx, sz) : out;
    if (!I || !Cm || !Y || !Z || !kvec || !o) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    hipLaunchKernelGGL(airy_kernel, dim3((P + 255) / 256, P), dim3(256), 0, ctx->stream, npad, P, ldp, obsc, I);
    hipLaunchKernelGGL(airy_filter_kernel, dim3(npad), dim3(256), 0, ctx->stream, npad, sigma, tophat_conv, kvec);
    hipLaunchKernelGGL(circulant_kernel, dim3((P + 255) / 256, P), dim3(256), 0, ctx->stream, kvec, npad, P, Cm);
    IMCOM_TRY(check_launch("airy setup"));
    IMCOM_TRY(launch_gemm(ctx, false, false, P, P, P, 1, I, P, 0, Cm, P, 0, Y, P, 0, 1.0, 0.0));  // Y = I C^T
    IMCOM_TRY(launch_gemm(ctx, false, true, P, P, P, 1, Cm, P, 0, Y, P, 0, Z, P, 0, 1.0, 0.0));   // Z = C Y
    hipLaunchKernelGGL(crop_kernel, dim3((n + 255) / 256, n), dim3(256), 0, ctx->stream, Z, P, kp, n, o);
    IMCOM_TRY(check_launch("crop_kernel"));
    if (host) {
        IMCOM_HIP_CHECK(hipMemcpyAsync(out, o, sz, hipMemcpyDeviceToHost, ctx->stream));
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return IMCOM_OK;
}

extern "C" int imcom_smooth_pad_width(double tophatwidth, double gaussiansigma)
{
    int npad = (int)ceil(tophatwidth + 6.0 * gaussiansigma + 1.0);  // coadd.py:453-454
    npad += ((4 - npad % 4) % 4);
    return npad;
}

// InImage.smooth_and_pad (coadd.py:433-474).  The reference multiplies the 2-D DFT of the padded image by the real, even,
// separable filter sinc(ux w) sinc(uy w) exp(-2 pi^2 s^2 (ux^2 + uy^2)) and keeps the real part of the inverse: a
// circular convolution along each axis with k[d] = (1/N) sum_u h(u) cos(2 pi u d / N), i.e. out = Cy I Cx^T with
// symmetric circulants -- two products on the fp64 MFMA GEMM for any image size, exact twiddles.
extern "C" int imcom_smooth_and_pad(imcom_ctx *ctx, int n, const double *in, int ny, int nx, double tophatwidth, double gaussiansigma,
                                    double *out, int memspace)
{
    IMCOM_TRY(ctx_ok3(ctx));
    IMCOM_REQUIRE(n >= 1 && in && out && ny >= 1 && nx >= 1 && tophatwidth >= 0.0 && gaussiansigma >= 0.0, "bad arguments");
    const bool host = memspace == IMCOM_MEM_HOST;
    const int npad = imcom_smooth_pad_width(tophatwidth, gaussiansigma), nyy = ny + 2 * npad, nxx = nx + 2 * npad;
    const int Py = (int)align_up((size_t)nyy, NB), Px = (int)align_up((size_t)nxx, NB);
    const size_t szI = (size_t)n * Py * Px * 8, szIn = (size_t)n * ny * nx * 8, szOut = (size_t)n * nyy * nxx * 8;
    IMCOM_TRY(ws_reserve(ctx, 3 * szI + (size_t)Py * Py * 8 + (size_t)Px * Px * 8 + (size_t)(nyy + nxx) * 8 + (host ? szIn + szOut : 0) + 16384));
    double *I = (double *)ws_take(ctx, szI), *Y = (double *)ws_take(ctx, szI), *Z = (double *)ws_take(ctx, szI);
    double *Cy = (double *)ws_take(ctx, (size_t)Py * Py * 8), *Cx = (double *)ws_take(ctx, (size_t)Px * Px * 8);
    double *ky = (double *)ws_take(ctx, (size_t)nyy * 8), *kx = (double *)ws_take(ctx, (size_t)nxx * 8);
    const double *src = in;
    double *dst = out;
    if (host) {
        double *in_d = (double *)ws_take(ctx, szIn);
        dst = (double *)ws_take(ctx, szOut);
        if (!in_d || !dst) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
        IMCOM_HIP_CHECK(hipMemcpyAsync(in_d, in, szIn, hipMemcpyHostToDevice, ctx->stream));
        src = in_d;
    }
    if (!I || !Y || !Z || !Cy || !Cx || !ky || !kx) { set_error("internal: workspace"); return IMCOM_ERR_NOMEM; }
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(pad_rect_kernel, dim3((Px + 255) / 256, Py, n), dim3(256), 0, st, src, ny, nx, npad, Py, Px, I);
    hipLaunchKernelGGL(airy_filter_kernel, dim3(nyy), dim3(256), 0, st, nyy, gaussiansigma, tophatwidth, ky);
    hipLaunchKernelGGL(airy_filter_kernel, dim3(nxx), dim3(256), 0, st, nxx, gaussiansigma, tophatwidth, kx);
    hipLaunchKernelGGL(circulant_kernel, dim3((Py + 255) / 256, Py), dim3(256), 0, st, ky, nyy, Py, Cy);
    hipLaunchKernelGGL(circulant_kernel, dim3((Px + 255) / 256, Px), dim3(256), 0, st, kx, nxx, Px, Cx);
    IMCOM_TRY(check_launch("smooth_and_pad setup"));
    IMCOM_TRY(launch_gemm(ctx, false, false, Py, Px, Px, n, I, Px, (long)Py * Px, Cx, Px, 0, Y, Px, (long)Py * Px, 1.0, 0.0));  // Y = I Cx^T
    IMCOM_TRY(launch_gemm(ctx, false, true, Py, Px, Py, n, Cy, Py, 0, Y, Px, (long)Py * Px, Z, Px, (long)Py * Px, 1.0, 0.0));   // Z = Cy Y
    hipLaunchKernelGGL(crop_rect_kernel, dim3((nxx + 255) / 256, nyy, n), dim3(256), 0, st, Z, Py, Px, nyy, nxx, dst);
    IMCOM_TRY(check_launch("crop_rect_kernel"));
    if (host) {
        IMCOM_HIP_CHECK(hipMemcpyAsync(out, dst, szOut, hipMemcpyDeviceToHost, st));
        IMCOM_HIP_CHECK(hipStreamSynchronize(st));
    }
    return IMCOM_OK;
}
