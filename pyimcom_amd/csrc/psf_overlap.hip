// psf_overlap.hip -- PSF cross-correlation tables on the device.
//
// Replaces PSFGrp.accel_pad_and_rfft2 (reference src/pyimcom/psfutil.py:943-986) and
// PSFOvl._build_psfovl / accel_irfft2_and_extract (1244-1294, 1178-1242):
//     table[p,q] = roll(irfft2(rfft2(pad(psf1[p])) * conj(rfft2(pad(psf2[q])))), nc)[:nsamp, :nsamp]
// written with the 6-pixel zero border the interpolators expect.
//
// Two formulations.  (1) Butterfly path (default whenever nfft <= 1024 is a product of 16, 8, 4, 2, 3, 5): line FFTs with
// one wavefront per line, butterflies in registers, the line exchanged through the wave's LDS slice without workgroup
// barriers (fft_lines.h), see below.
// (2) Dense-DFT path (general fallback, IMCOM_PSF_OVERLAP=gemm): the zero-padded 2-D DFTs as dense real matrix
// products with exact twiddle matrices (integer argument reduction mod nfft, then cos/sin of a multiple of
// pi/nfft) on the fp64 MFMA tile engine of gemm_f64.hip; only the kept nsamp x nsamp window of the inverse is
// computed; O(nfft * nsamp * nh) per stage, 51 us per cfg-2 table.  Both agree with numpy's FFTs to ~1e-15.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "fft_lines.h"
#include "launchers.h"

namespace imcom {

// twiddle(k, n) = (cos, sin)(2 pi k / nfft) with k reduced mod nfft exactly
__device__ __forceinline__ void twiddle(long k, int nfft, double *c, double *s)
{
    long r = k % nfft;
    if (r < 0) r += nfft;
    // exact special angles
    if (r == 0) { *c = 1.0; *s = 0.0; return; }
    if (2 * r == nfft) { *c = -1.0; *s = 0.0; return; }
    if (4 * r == nfft) { *c = 0.0; *s = 1.0; return; }
    if (4 * r == 3L * nfft) { *c = 0.0; *s = -1.0; return; }
    const double x = 2.0 * (double)r / (double)nfft;  // angle / pi in (0, 2)
    *c = cospi(x);
    *s = sinpi(x);
}

// kind 0: FX  [Kp x Np]  rows c (< nsamp), cols kx (< nh):   cos / -sin (2 pi c kx / nfft)
// kind 1: FY  [Mp x Kp]  rows ky (< nfft), cols r (< nsamp): cos / -sin (2 pi ky r / nfft)
// kind 2: IY  [Mp x Kp]  rows y' (< nsamp), cols ky (< nfft): cos / +sin (2 pi ky (y'-nc) / nfft)
// kind 3: IX  [Kp x Np]  rows kx (< nh), cols x' (< nsamp):  w cos / -w sin (2 pi kx (x'-nc) / nfft) / nfft^2
__global__ void dft_matrix_kernel(int kind, int rows, int cols, int rows_valid, int cols_valid, int nfft, int nc,
                                  double *__restrict__ Mc, double *__restrict__ Ms)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= cols || i >= rows) return;
    double c = 0.0, s = 0.0;
    if (i < rows_valid && j < cols_valid) {
        switch (kind) {
            case 0: twiddle((long)i * j, nfft, &c, &s); s = -s; break;
            case 1: twiddle((long)i * j, nfft, &c, &s); s = -s; break;
            case 2: twiddle((long)j * (i - nc), nfft, &c, &s); break;
            default: {
                twiddle((long)i * (j - nc), nfft, &c, &s);
                const double w = ((i == 0 || 2 * i == nfft) ? 1.0 : 2.0) / ((double)nfft * (double)nfft);
                c *= w;
                s *= -w;
            }
        }
    }
    Mc[(long)i * cols + j] = c;
    Ms[(long)i * cols + j] = s;
}

// zero-padded copy [n][ns][ns] -> [n][Kp][Kp]
__global__ void pad_psf_kernel(const double *__restrict__ psf, int ns, double *__restrict__ out, int Kp)
{
    const int p = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Kp) return;
    out[((long)p * Kp + r) * Kp + c] = (r < ns && c < ns) ? psf[((long)p * ns + r) * ns + c] : 0.0;
}

// Z[pair] = R1[p] * conj(R2[q]);   R layout [psf][2][Mp*Np]
// amp0 != 0: both spectra carry the Fourier-mode reweighting of PSFGrp.__init__ (psfutil.py:661-671),
// w(u) = 1 + amp0 exp(-2 pi^2 |u|^2 amps^2) with u = k / nfft wrapped to (-1/2, 1/2]; the product carries w^2.
__global__ void cmul_conj_kernel(const double *__restrict__ R1, const double *__restrict__ R2, const int *__restrict__ pairs,
                                 long plane, int Hp, int nfft, double amp0, double amps, double *__restrict__ Z)
{
    const int t = blockIdx.y;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const int p = pairs[2 * t], q = pairs[2 * t + 1];
    const double ar = R1[(2L * p) * plane + i], ai = R1[(2L * p + 1) * plane + i];
    const double br = R2[(2L * q) * plane + i], bi = R2[(2L * q + 1) * plane + i];
    double w2 = 1.0;
    if (amp0 != 0.0) {
        const int ky = (int)(i / Hp), kx = (int)(i % Hp);
        double uy = (double)ky / (double)nfft, ux = (double)kx / (double)nfft;
        if (uy > 0.5) uy -= 1.0;
        if (ux > 0.5) ux -= 1.0;
        const double w = 1.0 + amp0 * exp(-2.0 * M_PI * M_PI * (ux * ux + uy * uy) * (amps * amps));
        w2 = w * w;
    }
    Z[(2L * t) * plane + i] = (ar * br + ai * bi) * w2;
    Z[(2L * t + 1) * plane + i] = (ai * br - ar * bi) * w2;
}

// tables[t][6 + r][6 + c] = win[t][r][c] for r,c < ns; border zero
__global__ void crop_table_kernel(const double *__restrict__ win, int Wp, int ns, double *__restrict__ tables)
{
    const int ng = ns + 12;
    const int t = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ng) return;
    const int rr = r - 6, cc = c - 6;
    double v = 0.0;
    if (rr >= 0 && rr < ns && cc >= 0 && cc < ns) v = win[((long)t * Wp + rr) * Wp + cc];
    tables[((long)t * ng + r) * ng + c] = v;
}


// ---------------------------------------------------------------------------------------------------------
// Butterfly path: line FFTs with one wavefront per line (fft_lines.h), used whenever nfft <= 1024 factors into 16, 8, 4,
// 2, 3, 5; the dense-DFT GEMM form above stays as the general fallback and as an independent cross-check
// (IMCOM_PSF_OVERLAP=gemm).
//   forward:  rows two-for-one (two real PSF rows ride as one complex line), then the columns of the half spectrum
//   inverse:  columns of R1 conj(R2) (the product is formed in the first stage's loads), kept rows only (stored by the
//             last stage, rolled by nc); then rows two-for-one from the Hermitian half back to two real window rows,
//             rolled by nc, cropped and scaled on store, inside the 6-sample zero border the same kernel writes.
// Workgroups are ordered pair-fastest: the workgroups in flight work on the same few spectrum columns of different
// pairs, so a spectrum column is fetched from HBM once per L2 instead of once per table that uses it.
__global__ void fft_twiddle_kernel(FftPlan pl, cplx *__restrict__ tw)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= pl.twn) return;
    int s = 1, Ns = pl.radix[0];
    while (s + 1 < pl.nst && e >= pl.twoff[s + 1]) { Ns *= pl.radix[s]; s++; }
    const int local = e - pl.twoff[s], t = local / Ns + 1, k = local - (t - 1) * Ns;
    double c, sn;
    twiddle((long)t * k * (pl.n / (Ns * pl.radix[s])), pl.n, &c, &sn);
    tw[e] = make_double2(c, -sn);
}

// V, the intermediate between the two inverse transforms, is [pair][row pair l][kx][2]: rows 2l, 2l+1 of a column next to each
// other (the row transform takes them as one complex line and reads its line as one contiguous run), columns padded to a
// multiple of 4 so that four neighbouring columns of a row pair are exactly one aligned 128-byte line.
__host__ __device__ constexpr int v_stride(int n) { return (n / 2 + 1 + 3) / 4 * 4; }

#define IMCOM_WF_PROLOGUE                                                                        \
    extern __shared__ cplx fbuf[];                                                               \
    const int n = pl.n, nh = n / 2 + 1, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  \
    cplx *twl = fbuf + pl.waves * pl.npad; /* the stage tables ride in LDS behind the lines */   \
    for (int e = threadIdx.x; e < pl.twn; e += blockDim.x) twl[e] = tw[e];                       \
    __syncthreads();                                                                             \
    cplx *line = fbuf + wave * pl.npad

// forward, along x: rows 2l, 2l+1 of PSF p as one complex line; Y1[p][kx][row], kx < nh
__global__ __launch_bounds__(WF_MAXWAVES * 64) void fft_fwd_rows_kernel(const double *__restrict__ psf, int ns, FftPlan pl,
                                                            const cplx *__restrict__ tw, cplx *__restrict__ Y1)
{
    IMCOM_WF_PROLOGUE;
    const int p = blockIdx.x, r0 = 2 * (blockIdx.y * pl.waves + wave), lane = threadIdx.x & 63;
    if (r0 >= ns) return;
    const double *row0 = psf + ((long)p * ns + r0) * ns, *row1 = row0 + ns;
    const bool two = r0 + 1 < ns;
    auto load0 = [&](int x) { return x < ns ? make_double2(row0[x], two ? row1[x] : 0.0) : make_double2(0.0, 0.0); };
    auto keep = [line](int i, cplx v) { line[wf_swz(i)] = v; };
    wf_line<false>(line, twl, pl, load0, keep);
    // z = FFT(a + i b) of two real rows: A_k = (z_k + conj z_{n-k}) / 2, B_k = (z_k - conj z_{n-k}) / (2 i)
    for (int k = lane; k < nh; k += 64) {
        const cplx zk = line[wf_swz(k)], zm = line[wf_swz(k ? n - k : 0)];
        cplx *dst = Y1 + ((long)p * nh + k) * ns + r0;
        dst[0] = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        if (two) dst[1] = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
    }
}

// forward, along y: column kx of Y1[p] (rows >= ns are zero) -> R[p][kx][ky]
__global__ __launch_bounds__(WF_MAXWAVES * 64) void fft_fwd_cols_kernel(const cplx *__restrict__ Y1, int ns, FftPlan pl,
                                                            const cplx *__restrict__ tw, cplx *__restrict__ R)
{
    IMCOM_WF_PROLOGUE;
    const int p = blockIdx.x, kx = blockIdx.y * pl.waves + wave;
    if (kx >= nh) return;
    const cplx *src = Y1 + ((long)p * nh + kx) * ns;
    cplx *dst = R + ((long)p * nh + kx) * n;
    auto load0 = [&](int y) { return y < ns ? src[y] : make_double2(0.0, 0.0); };
    auto storeN = [&](int ky, cplx v) { dst[ky] = v; };
    wf_line<false>(line, twl, pl, load0, storeN);
}

// inverse, along y: column kx of R1[p] conj(R2[q]) (x the squared Fourier-mode weight) -> V[t][y' / 2][kx][y' & 1] for the
// kept rows y' < ns (source row (y' - nc) mod n: the roll of psfutil.py:1225-1232).  Rows 2l, 2l+1 sit next to each other
// because the row transform takes them as one complex line: it then reads its line as one contiguous run.
__global__ __launch_bounds__(WF_MAXWAVES * 64) void fft_inv_cols_kernel(const cplx *__restrict__ Ra, const cplx *__restrict__ Rb,
                                                            const int *__restrict__ pairs, int ns, FftPlan pl,
                                                            const cplx *__restrict__ tw, double amp0, double amps,
                                                            cplx *__restrict__ V)
{
    IMCOM_WF_PROLOGUE;
    const int t = blockIdx.x, kx = blockIdx.y * pl.waves + wave, nc = ns / 2;
    if (kx >= nh) return;
    const cplx *R1 = Ra + ((long)pairs[2 * t] * nh + kx) * n, *R2 = Rb + ((long)pairs[2 * t + 1] * nh + kx) * n;
    const int nhp = v_stride(n);
    cplx *dst = V + ((long)t * ((ns + 1) / 2) * nhp + kx) * 2;
    double ux = (double)kx / (double)n;
    if (ux > 0.5) ux -= 1.0;
    auto load0 = [&](int ky) {
        const cplx a = R1[ky], b = R2[ky];
        double w2 = 1.0;
        if (amp0 != 0.0) {
            double uy = (double)ky / (double)n;
            if (uy > 0.5) uy -= 1.0;
            const double w = 1.0 + amp0 * exp(-2.0 * M_PI * M_PI * (ux * ux + uy * uy) * (amps * amps));
            w2 = w * w;
        }
        return make_double2((a.x * b.x + a.y * b.y) * w2, (a.y * b.x - a.x * b.y) * w2);
    };
    auto storeN = [&](int i, cplx v) {
        const int yp = i + nc - (i + nc >= n ? n : 0);
        if (yp < ns) dst[(yp >> 1) * (2 * nhp) + (yp & 1)] = v;
    };
    wf_line<true>(line, twl, pl, load0, storeN);
}

// inverse, along x: window rows 2l, 2l+1 of pair t from their Hermitian halves as one complex line; the real and the
// imaginary part of the result are the two rows.  numpy's c2r ignores the imaginary parts of the DC and Nyquist
// bins; so does this.  Stored rolled by nc, cropped to ns, scaled by 1/n^2, inside the 6-sample zero border (the
// waves of a row pair write the side borders of their rows, the first and the last pair also the rows above / below).
__global__ __launch_bounds__(WF_MAXWAVES * 64) void fft_inv_rows_kernel(const cplx *__restrict__ V, int ns, FftPlan pl,
                                                            const cplx *__restrict__ tw, const int *__restrict__ slot,
                                                            double *__restrict__ tables)
{
    IMCOM_WF_PROLOGUE;
    const int t = blockIdx.x, r0 = 2 * (blockIdx.y * pl.waves + wave), nc = ns / 2, ng = ns + 12, lane = threadIdx.x & 63;
    if (r0 >= ns) return;
    const bool two = r0 + 1 < ns;
    const cplx *src = V + ((long)t * ((ns + 1) / 2) + r0 / 2) * v_stride(n) * 2;
    double *tab = tables + (long)(slot ? slot[t] : t) * ng * ng;  // slot: the arena table this pair's result goes to
    const int bo = two ? 1 : 0;
    const double bz = two ? 1.0 : 0.0;
    auto load0 = [&](int k) {
        const int kk = k < nh ? k : n - k;
        const cplx *s2 = src + 2 * kk;
        const cplx a = s2[0], b = s2[bo];  // branch free: a lone last row reads itself and is weighted by zero
        const double sg = (k >= nh) ? -1.0 : ((kk == 0 || 2 * kk == n) ? 0.0 : 1.0);  // conjugate half; real DC / Nyquist bins
        return make_double2(a.x - sg * bz * b.y, sg * a.y + bz * b.x);
    };
    const double scale = 1.0 / ((double)n * (double)n);
    double *o0 = tab + (long)(6 + r0) * ng + 6, *o1 = o0 + ng;
    auto storeN = [&](int i, cplx z) {
        const int xp = i + nc - (i + nc >= n ? n : 0);
        if (xp < ns) {
            o0[xp] = z.x * scale;
            if (two) o1[xp] = z.y * scale;
        }
    };
    wf_line<true>(line, twl, pl, load0, storeN);
    // zero border
    if (lane < 12) {
        const int c = lane < 6 ? lane - 6 : ns + lane - 6;
        o0[c] = 0.0;
        if (two) o1[c] = 0.0;
    }
    if (r0 == 0)
        for (int e = lane; e < 6 * ng; e += 64) tab[e] = 0.0;
    if (r0 + 2 >= ns) {
        double *bot = tab + (long)(6 + ns) * ng;
        for (int e = lane; e < 6 * ng; e += 64) bot[e] = 0.0;
    }
}

// nfft = product of radices 16, 8, 4, 2, 3, 5 (at least two stages) with lines that fit the wave engine?
static bool fft_plan(int n, FftPlan *pl)
{
    if (n > WF_MAXN || n < 6) return false;
    pl->n = n;
    pl->npad = (n + 15) / 16 * 16;
    pl->nst = 0;
    int r = n;
    const int cand[6] = {16, 8, 4, 2, 3, 5};
    for (int ci = 0; ci < 6; ci++)
        while (r % cand[ci] == 0) {
            if (pl->nst >= WF_MAXST) return false;
            pl->radix[pl->nst++] = cand[ci];
            r /= cand[ci];
        }
    if (r != 1 || pl->nst < 2) return false;
    int Ns = pl->radix[0], off = 0;
    pl->twoff[0] = 0;
    for (int s = 1; s < pl->nst; s++) {
        pl->twoff[s] = off;
        off += (pl->radix[s] - 1) * Ns;
        Ns *= pl->radix[s];
    }
    pl->twn = off;
    // as many waves (= lines) per workgroup as LDS holds next to the tables, WF_MAXWAVES at most
    const long line = (long)pl->npad * 16, lds = 160 * 1024;
    pl->waves = (int)std::min<long>(WF_MAXWAVES, (lds - (long)pl->twn * 16) / line);
    if (const char *e = getenv("IMCOM_FFT_WAVES")) pl->waves = std::max(1, std::min(pl->waves, atoi(e)));  // tuning runs
    return pl->waves >= 1;
}

static bool fft_force_gemm()
{
    static const bool f = getenv("IMCOM_PSF_OVERLAP") && !strcmp(getenv("IMCOM_PSF_OVERLAP"), "gemm");
    return f;
}

static size_t fft_lds_bytes(const FftPlan &pl) { return ((size_t)pl.waves * pl.npad + pl.twn) * 16; }

static int fft_set_lds(const FftPlan &pl)
{
    const size_t lds = fft_lds_bytes(pl);
    if (lds <= 48 * 1024) return IMCOM_OK;
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)fft_fwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)fft_fwd_cols_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)fft_inv_cols_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)fft_inv_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return IMCOM_OK;
}

// ---- static shape n = 256 r: persistent workgroups (one per CU), every wave loops over lines; the line index runs
// pair-fastest for the column transforms (see above), row-fastest for the row transforms -------------------------------------------------
#define IMCOM_WF16_PROLOGUE                                                                      \
    extern __shared__ cplx fbuf[];                                                               \
    constexpr int n = Wf16<R2>::N, nh = n / 2 + 1;                                               \
    const int W = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63; \
    cplx *twl = fbuf + W * Wf16<R2>::LINE;                                                       \
    for (int e = threadIdx.x; e < Wf16<R2>::TWN; e += blockDim.x) twl[e] = tw[e];                \
    __syncthreads();                                                                             \
    cplx *line = fbuf + wave * Wf16<R2>::LINE;                                                   \
    (void)lane

template <int R2>
__global__ __launch_bounds__(WF_MAXWAVES * 64) void wf16_fwd_rows_kernel(const double *__restrict__ psf, int npsf, int ns,
                                                                         const cplx *__restrict__ tw, cplx *__restrict__ Y1)
{
    IMCOM_WF16_PROLOGUE;
    const long total = (long)npsf * ((ns + 1) / 2);
    for (long L = (long)blockIdx.x * W + wave; L < total; L += (long)gridDim.x * W) {
        const int rp = (ns + 1) / 2, p = (int)(L / rp), r0 = 2 * (int)(L % rp);  // neighbours write neighbouring pieces of Y1's lines
        const double *row0 = psf + ((long)p * ns + r0) * ns, *row1 = row0 + ns;
        const bool two = r0 + 1 < ns;
        auto load0 = [&](int x) { return x < ns ? make_double2(row0[x], two ? row1[x] : 0.0) : make_double2(0.0, 0.0); };
        auto keep = [line](int i, cplx v) { line[wf_pad16(i)] = v; };
        wf16_line<R2, false>(line, twl, load0, keep);
        // z = FFT(a + i b) of two real rows: A_k = (z_k + conj z_{n-k}) / 2, B_k = (z_k - conj z_{n-k}) / (2 i)
        for (int k = lane; k < nh; k += 64) {
            const cplx zk = line[wf_pad16(k)], zm = line[wf_pad16(k ? n - k : 0)];
            cplx *dst = Y1 + ((long)p * nh + k) * ns + r0;
            dst[0] = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
            if (two) dst[1] = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int R2>
__global__ __launch_bounds__(WF_MAXWAVES * 64) void wf16_fwd_cols_kernel(const cplx *__restrict__ Y1, int npsf, int ns,
                                                                         const cplx *__restrict__ tw, cplx *__restrict__ R)
{
    IMCOM_WF16_PROLOGUE;
    const long total = (long)npsf * nh;
    for (long L = (long)blockIdx.x * W + wave; L < total; L += (long)gridDim.x * W) {
        const int p = (int)(L % npsf), kx = (int)(L / npsf);
        const cplx *src = Y1 + ((long)p * nh + kx) * ns;
        cplx *dst = R + ((long)p * nh + kx) * n;
        auto load0 = [&](int y) { return y < ns ? src[y] : make_double2(0.0, 0.0); };
        auto storeN = [&](int ky, cplx v) { dst[ky] = v; };
        wf16_line<R2, false>(line, twl, load0, storeN);
    }
}

constexpr int WF_COLS_WAVES = 8;  // waves per workgroup of the column kernel: two per SIMD, 256 registers each
template <int R2, bool AMP>
__global__ __launch_bounds__(WF_COLS_WAVES * 64) void wf16_inv_cols_kernel(const cplx *__restrict__ Ra, const cplx *__restrict__ Rb,
                                                                         const int *__restrict__ pairs, int npairs, int ns,
                                                                         const cplx *__restrict__ tw, double amp0, double amps,
                                                                         const int *__restrict__ win, cplx *__restrict__ V)
{
    // (no copy of the stage tables in LDS: this kernel runs two waves per SIMD -- eight lines fill the LDS -- and a lane's twiddles, the
    // same for every line, stay in 92 of its 256 registers: the tables' LDS reads were a tenth of the kernel, tools/ab_fft_abl.sh)
    extern __shared__ cplx fbuf[];
    constexpr int n = Wf16<R2>::N, nh = n / 2 + 1;
    const int W = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    cplx *line = fbuf + wave * Wf16<R2>::LINE;
    WfTw<R2> twr;
    wf16_load_tw<R2>(twr, tw);
    const int nc = ns / 2, rp = (ns + 1) / 2;
    constexpr int nhp = v_stride(n);
    // Line order: four neighbouring columns of a pair, then the next pair, ...  Workgroups are dealt round-robin over the 8 XCDs
    // (id % 8), each with its own L2: XCD x takes the pairs [p0, p1) -- an eighth of the list, in which the pairs of a table set
    // (which share their 6-12 spectra) are neighbours -- so a spectrum column is fetched into ONE L2 and reused there by every
    // pair of the set (the grid is a multiple of 8 workgroups).
    // The workgroup (a multiple of four waves) works in rounds: every wave transforms one column and leaves the result in its
    // LDS line; then each group of four waves -- four neighbouring columns of one pair -- writes V together, 128-byte line by
    // line (a wave alone would scatter 32-byte pieces 12 KB apart: 1.5x the kernel's time).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int p0 = (int)((long)npairs * xcd / 8), np = (int)((long)npairs * (xcd + 1) / 8) - p0;
    const long total = 4L * np * ((nh + 3) / 4), per_round = (long)nslot * W;
    for (long base = (long)slot * W; base < total; base += per_round) {  // base is a multiple of 4: uniform over the workgroup
        const long L = base + wave;
        const int g = (int)(L / (4L * np)), rem = (int)(L - 4L * np * g), t = p0 + (rem >> 2), kx = 4 * g + (rem & 3);
        if (L < total && kx < nh) {
            const cplx *R1 = Ra + ((long)pairs[2 * t] * nh + kx) * n, *R2p = Rb + ((long)pairs[2 * t + 1] * nh + kx) * n;
            double ux = (double)kx / (double)n;
            if (ux > 0.5) ux -= 1.0;
            auto load0 = [&](int ky) {
                const cplx a = R1[ky], b = R2p[ky];
                cplx z = make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
                if (AMP) {
                    double uy = (double)ky / (double)n;
                    if (uy > 0.5) uy -= 1.0;
                    const double w = 1.0 + amp0 * exp(-2.0 * M_PI * M_PI * (ux * ux + uy * uy) * (amps * amps));
                    z.x *= w * w;
                    z.y *= w * w;
                }
                return z;
            };
            auto keep = [line](int i, cplx v) { line[wf_pad16(i)] = v; };
            wf16_line_regs<R2, true>(line, twr, load0, keep);
        }
        __syncthreads();
        // group q = wave >> 2 holds columns 4g .. 4g+3 of pair t in lines 4q .. 4q+3; lane = (row pair within a block of 8, column,
        // row of the pair): one store instruction writes 8 whole 128-byte lines.  Row y' of the window is output (y' - nc) mod n.
        if (L - (wave & 3) < total) {  // the group's first line exists (then so do g and t, the same for its four waves)
            const int ncol = min(4, nh - 4 * g), c2 = (lane >> 1) & 3, h = lane & 1;
            const cplx *src = fbuf + ((wave & ~3) + c2) * Wf16<R2>::LINE;
            cplx *dst = V + ((long)t * rp * nhp + 4 * g) * 2 + (lane & 7);
            // window rows the caller will read (win: [row_lo, row_hi) of the pair's table window; the rest of V stays unwritten)
            const int llo = win ? win[4 * t] >> 1 : 0, lhi = win ? (win[4 * t + 1] + 1) >> 1 : rp;
            for (int l = 8 * (wave & 3) + (lane >> 3); l < rp; l += 32) {
                const int i = 2 * l + h - nc + (2 * l + h < nc ? n : 0);
                if (c2 < ncol && l >= llo && l < lhi) {
                    const cplx v = src[wf_pad16(i)];
                    if (!(IMCOM_FFT_ABL & 2) || v.x == 1.2345e300) dst[(long)l * (2 * nhp)] = v;
                }
            }
        }
        __syncthreads();
    }
}

template <int R2>
__global__ __launch_bounds__(WF_MAXWAVES * 64) void wf16_inv_rows_kernel(const cplx *__restrict__ V, int npairs, int ns,
                                                                         const cplx *__restrict__ tw, const int *__restrict__ win,
                                                                         const int *__restrict__ slot, double *__restrict__ tables)
{
    IMCOM_WF16_PROLOGUE;
    const int nc = ns / 2, ng = ns + 12;
    const double scale = 1.0 / ((double)n * (double)n);
    const long total = (long)npairs * ((ns + 1) / 2);
    for (long L = (long)blockIdx.x * W + wave; L < total; L += (long)gridDim.x * W) {
        const int r0 = 2 * (int)(L % ((ns + 1) / 2));  // V is [pair][row pair][kx][2]: line L is one contiguous run
        const int t = (int)(L / ((ns + 1) / 2));
        // win: only rows [w0, w1) and columns [c0, c1) of this pair's window are ever read (cross tables of two PSF groups:
        // the separations between their pixels have one sign); the rest of the table is left as it is
        int c0 = 0, c1 = ns;
        if (win) {
            const int w0 = __builtin_amdgcn_readfirstlane(win[4 * t]), w1 = __builtin_amdgcn_readfirstlane(win[4 * t + 1]);
            if (r0 + 1 < w0 || r0 >= w1) continue;
            c0 = win[4 * t + 2];
            c1 = win[4 * t + 3];
        }
        const bool two = r0 + 1 < ns;
        const cplx *src = V + L * v_stride(n) * 2;
        double *tab = tables + (long)(slot ? __builtin_amdgcn_readfirstlane(slot[t]) : t) * ng * ng;  // slot: the arena table of this pair
        const int bo = two ? 1 : 0;
        const double bz = two ? 1.0 : 0.0;
        auto load0 = [&](int k) {
            const int kk = k < nh ? k : n - k;
            const cplx *s2 = src + 2 * kk;
            const cplx a = s2[0], b = s2[bo];  // branch free: a lone last row reads itself and is weighted by zero
            const double sg = (k >= nh) ? -1.0 : ((kk == 0 || 2 * kk == n) ? 0.0 : 1.0);  // conjugate half; real DC / Nyquist bins
            return make_double2(a.x - sg * bz * b.y, sg * a.y + bz * b.x);
        };
        double *o0 = tab + (long)(6 + r0) * ng + 6, *o1 = o0 + ng;
        auto storeN = [&](int i, cplx z) {
            const int xp = i + nc - (i + nc >= n ? n : 0);
            if (xp < c1 && xp >= c0) {
                o0[xp] = z.x * scale;
                if (two) o1[xp] = z.y * scale;
            }
        };
        wf16_line<R2, true>(line, twl, load0, storeN);
        if (lane < 12) {  // zero border
            const int c = lane < 6 ? lane - 6 : ns + lane - 6;
            o0[c] = 0.0;
            if (two) o1[c] = 0.0;
        }
        if (r0 == 0)
            for (int e = lane; e < 6 * ng; e += 64) tab[e] = 0.0;
        if (r0 + 2 >= ns) {
            double *bot = tab + (long)(6 + ns) * ng;
            for (int e = lane; e < 6 * ng; e += 64) bot[e] = 0.0;
        }
    }
}

// n = 256 r with the plan {16, 16, r}?
static int fft_static_r(const FftPlan &pl)
{
    static const bool off = getenv("IMCOM_FFT_GENERIC") != nullptr;  // A/B and test runs: the general kernels for every n
    if (off || pl.nst != 3 || pl.radix[0] != 16 || pl.radix[1] != 16) return 0;
    return pl.radix[2] >= 2 && pl.radix[2] <= 4 ? pl.radix[2] : 0;
}
template <int R2> static int wf16_waves()
{
    int w = (int)std::min<long>(WF_MAXWAVES, (160L * 1024 - Wf16<R2>::TWN * 16L) / (Wf16<R2>::LINE * 16L));
    if (const char *e = getenv("IMCOM_FFT_WAVES")) w = std::max(4, std::min(w, atoi(e)));  // tuning runs
    return w;
}
template <int R2> static size_t wf16_lds(int waves) { return ((size_t)waves * Wf16<R2>::LINE + Wf16<R2>::TWN) * 16; }

template <int R2>
static int wf16_forward(imcom_ctx *ctx, const double *psf, int npsf, int nsamp, const cplx *tw, cplx *Y1, cplx *R)
{
    const int W = wf16_waves<R2>(), nh = Wf16<R2>::N / 2 + 1;
    const size_t lds = wf16_lds<R2>(W);
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)wf16_fwd_rows_kernel<R2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)wf16_fwd_cols_kernel<R2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long rows = (long)npsf * ((nsamp + 1) / 2), cols = (long)npsf * nh;
    const int g1 = (int)std::min<long>(ctx->cu_count, (rows + W - 1) / W), g2 = (int)std::min<long>(ctx->cu_count, (cols + W - 1) / W);
    hipLaunchKernelGGL(wf16_fwd_rows_kernel<R2>, dim3(g1), dim3(64 * W), lds, ctx->stream, psf, npsf, nsamp, tw, Y1);
    hipLaunchKernelGGL(wf16_fwd_cols_kernel<R2>, dim3(g2), dim3(64 * W), lds, ctx->stream, (const cplx *)Y1, npsf, nsamp, tw, R);
    return check_launch("psf spectra (16 x 16 x r lines)");
}

template <int R2>
static int wf16_inverse(imcom_ctx *ctx, const cplx *Ra, const cplx *Rb, const int *pairs_dev, int npairs, int nsamp, const cplx *tw,
                        double a0, double a1, const int *win_dev, const int *slot_dev, cplx *V, double *tables)
{
    // the column kernel's waves write V in groups of four (four neighbouring columns = one 128-byte line per row pair)
    const int W = wf16_waves<R2>(), nh = Wf16<R2>::N / 2 + 1;
    // The column kernel's rounds -- transform, barrier, the groups' stores, barrier -- keep the waves of a workgroup in step.
    // IMCOM_FFT_COLS_SPLIT=1: workgroups of ONE group of four waves, as many per CU as the LDS holds (two at n = 768), each in a phase
    // of its own: 2.31 -> 2.22 us per table on a bare request of 3600 tables, nothing inside a block (1752 against 1754 ms per 48 x 48
    // block; profiles/r04_negative_results.txt item 7) -- not the default.
    static const bool split = getenv("IMCOM_FFT_COLS_SPLIT") && atoi(getenv("IMCOM_FFT_COLS_SPLIT")) > 0;
    const int Wc = split ? 4 : std::min(WF_COLS_WAVES, W / 4 * 4);
    const size_t lds = wf16_lds<R2>(W), ldsc = (size_t)Wc * Wf16<R2>::LINE * 16;  // (the column kernel keeps no stage tables in LDS)
    const int per_cu = split ? (int)std::max<size_t>(1, std::min<size_t>(WF_COLS_WAVES / 4, (160 * 1024) / ldsc)) : 1;
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)wf16_inv_cols_kernel<R2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc));
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)wf16_inv_cols_kernel<R2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc));
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)wf16_inv_rows_kernel<R2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long cols = (long)npairs * nh, rows = (long)npairs * ((nsamp + 1) / 2);
    const int g1 = (int)std::max<long>(8, std::min<long>((long)per_cu * ctx->cu_count, (cols + Wc - 1) / Wc) / 8 * 8);  // a multiple of the 8 XCDs
    const int g2 = (int)std::min<long>(ctx->cu_count, (rows + W - 1) / W);
    if (a0 != 0.0)
        hipLaunchKernelGGL((wf16_inv_cols_kernel<R2, true>), dim3(g1), dim3(64 * Wc), ldsc, ctx->stream, Ra, Rb, pairs_dev, npairs, nsamp, tw, a0, a1, win_dev, V);
    else
        hipLaunchKernelGGL((wf16_inv_cols_kernel<R2, false>), dim3(g1), dim3(64 * Wc), ldsc, ctx->stream, Ra, Rb, pairs_dev, npairs, nsamp, tw, a0, a1, win_dev, V);
    hipLaunchKernelGGL(wf16_inv_rows_kernel<R2>, dim3(g2), dim3(64 * W), lds, ctx->stream, (const cplx *)V, npairs, nsamp, tw, win_dev, slot_dev, tables);
    return check_launch("psf_overlap (16 x 16 x r lines)");
}

static size_t fft_forward_ws(int n, int nsamp, int nfft) { return (size_t)nfft * 16 + (size_t)n * nsamp * (nfft / 2 + 1) * 16 + 1024; }
static size_t fft_inverse_ws(int npairs, int nsamp, int nfft) { return (size_t)nfft * 16 + (size_t)npairs * (nsamp + 1) * v_stride(nfft) * 16 + (size_t)npairs * 28 + 4096; }

static cplx *fft_twiddles(imcom_ctx *ctx, const FftPlan &pl)
{
    cplx *tw = (cplx *)ws_take(ctx, (size_t)pl.n * 16);  // twn < n
    if (tw) hipLaunchKernelGGL(fft_twiddle_kernel, dim3((pl.twn + 255) / 256), dim3(256), 0, ctx->stream, pl, tw);
    return tw;
}

// spectra R[n][nh][nfft] (complex) of n sampled PSFs; the caller has reserved fft_forward_ws() of workspace
static int fft_forward(imcom_ctx *ctx, const FftPlan &pl, const double *psf, int n, int nsamp, cplx *R)
{
    const int nfft = pl.n, nh = nfft / 2 + 1;
    cplx *tw = fft_twiddles(ctx, pl);
    cplx *Y1 = (cplx *)ws_take(ctx, (size_t)n * nsamp * nh * 16);
    if (!tw || !Y1) { set_error("internal: workspace plan too small"); return IMCOM_ERR_NOMEM; }
    switch (fft_static_r(pl)) {
    case 2: return wf16_forward<2>(ctx, psf, n, nsamp, tw, Y1, R);
    case 3: return wf16_forward<3>(ctx, psf, n, nsamp, tw, Y1, R);
    case 4: return wf16_forward<4>(ctx, psf, n, nsamp, tw, Y1, R);
    default: break;
    }
    IMCOM_TRY(fft_set_lds(pl));
    hipStream_t st = ctx->stream;
    const size_t lds = fft_lds_bytes(pl);
    const int W = pl.waves, row_blocks = ((nsamp + 1) / 2 + W - 1) / W, col_blocks = (nh + W - 1) / W;
    hipLaunchKernelGGL(fft_fwd_rows_kernel, dim3(n, row_blocks), dim3(64 * W), lds, st, psf, nsamp, pl, tw, Y1);
    hipLaunchKernelGGL(fft_fwd_cols_kernel, dim3(n, col_blocks), dim3(64 * W), lds, st, Y1, nsamp, pl, tw, R);
    return check_launch("psf spectra (butterfly path)");
}

// tables[t] from spectra pairs (Ra[pairs[2t]], Rb[pairs[2t+1]]); the caller has reserved fft_inverse_ws()
static int fft_inverse(imcom_ctx *ctx, const FftPlan &pl, const cplx *Ra, const cplx *Rb, const int *pairs_host, int npairs,
                       int nsamp, const double *amp_penalty, double *tables, const int *win_host = nullptr, const int *slots_host = nullptr)
{
    const int nfft = pl.n, nh = nfft / 2 + 1;
    cplx *tw = fft_twiddles(ctx, pl);
    cplx *V = (cplx *)ws_take(ctx, (size_t)npairs * (nsamp + 1) * v_stride(nfft) * 16);  // [pair][row pair][kx, stride nhp][2]
    int *pairs_dev = (int *)ws_take(ctx, (size_t)npairs * 8);
    int *win_dev = win_host ? (int *)ws_take(ctx, (size_t)npairs * 16) : nullptr;  // the static kernels honour it; the general ones fill whole tables
    int *slot_dev = slots_host ? (int *)ws_take(ctx, (size_t)npairs * 4) : nullptr;
    if (!tw || !V || !pairs_dev || (win_host && !win_dev) || (slots_host && !slot_dev)) { set_error("internal: workspace plan too small"); return IMCOM_ERR_NOMEM; }
    IMCOM_TRY(fft_set_lds(pl));
    hipStream_t st = ctx->stream;
    IMCOM_TRY(upload(ctx, pairs_dev, pairs_host, 2 * (size_t)npairs));  // through the pinned ring: no stream drain
    if (win_host) IMCOM_TRY(upload(ctx, win_dev, win_host, 4 * (size_t)npairs));
    if (slots_host) IMCOM_TRY(upload(ctx, slot_dev, slots_host, (size_t)npairs));
    const double a0 = amp_penalty ? amp_penalty[0] : 0.0, a1 = amp_penalty ? amp_penalty[1] : 0.0;
    switch (fft_static_r(pl)) {
    case 2: return wf16_inverse<2>(ctx, Ra, Rb, pairs_dev, npairs, nsamp, tw, a0, a1, win_dev, slot_dev, V, tables);
    case 3: return wf16_inverse<3>(ctx, Ra, Rb, pairs_dev, npairs, nsamp, tw, a0, a1, win_dev, slot_dev, V, tables);
    case 4: return wf16_inverse<4>(ctx, Ra, Rb, pairs_dev, npairs, nsamp, tw, a0, a1, win_dev, slot_dev, V, tables);
    default: break;
    }
    const size_t lds = fft_lds_bytes(pl);
    const int W = pl.waves, row_blocks = ((nsamp + 1) / 2 + W - 1) / W, col_blocks = (nh + W - 1) / W;
    hipLaunchKernelGGL(fft_inv_cols_kernel, dim3(npairs, col_blocks), dim3(64 * W), lds, st, Ra, Rb, pairs_dev, nsamp, pl, tw, a0, a1, V);
    hipLaunchKernelGGL(fft_inv_rows_kernel, dim3(npairs, row_blocks), dim3(64 * W), lds, st, V, nsamp, pl, tw, slot_dev, tables);
    return check_launch("psf_overlap (butterfly path)");
}

static int up(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace imcom

using namespace imcom;

extern "C" int imcom_psf_overlap(imcom_ctx *ctx, const double *psf1, int n1, const double *psf2, int n2, int nsamp,
                                 int nfft, const int *pairs_host, int npairs, const double *amp_penalty, double *tables)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(psf1 && psf2 && pairs_host && tables, "null pointer");
    IMCOM_REQUIRE(n1 >= 1 && n2 >= 1 && npairs >= 1 && nsamp >= 1 && nfft >= 2 * nsamp && nfft % 2 == 0,
                  "bad sizes (need nfft even and >= 2*nsamp)");
    IMCOM_REQUIRE(nsamp % 2 == 1, "nsamp must be odd (PSFGrp.setup: nsamp = npixpsf*oversamp - 1)");
    for (int t = 0; t < npairs; t++)
        IMCOM_REQUIRE(pairs_host[2 * t] >= 0 && pairs_host[2 * t] < n1 && pairs_host[2 * t + 1] >= 0 && pairs_host[2 * t + 1] < n2,
                      "pair %d out of range", t);
    FftPlan pl;
    if (!fft_force_gemm() && fft_plan(nfft, &pl)) {
        const int nh_ = nfft / 2 + 1;
        const bool same_ = (psf1 == psf2 && n1 == n2);
        const int npsf_ = same_ ? n1 : n1 + n2;
        IMCOM_TRY(ws_reserve(ctx, fft_forward_ws(npsf_, nsamp, nfft) + (size_t)npsf_ * nfft * nh_ * 16 + fft_inverse_ws(npairs, nsamp, nfft) + (size_t)nfft * 64 + 65536));
        cplx *R = (cplx *)ws_take(ctx, (size_t)npsf_ * nfft * nh_ * 16);
        if (!R) { set_error("internal: workspace plan too small"); return IMCOM_ERR_NOMEM; }
        ProfScope ps(ctx, "psf_overlap");
        IMCOM_TRY(fft_forward(ctx, pl, psf1, n1, nsamp, R));
        if (!same_) IMCOM_TRY(fft_forward(ctx, pl, psf2, n2, nsamp, R + (long)n1 * nfft * nh_));
        return fft_inverse(ctx, pl, R, same_ ? R : R + (long)n1 * nfft * nh_, pairs_host, npairs, nsamp, amp_penalty, tables);
    }
    const int nh = nfft / 2 + 1, nc = nsamp / 2;
    const int Sp = up(nsamp, NB);  // padded nsamp (as an M/N extent and as a K extent)
    const int Hp = up(nh, NB);     // padded half-spectrum width
    const int Fp = up(nfft, NB);   // padded nfft
    const bool same = (psf1 == psf2 && n1 == n2);
    const long plane = (long)Fp * Hp;
    auto B8 = [](long n) { return (size_t)n * 8; };
    size_t total = 0;
    auto plan = [&](size_t b) { total = align_up(total, 256) + b; };
    plan(B8((long)Sp * Hp) * 2);        // FX
    plan(B8((long)Fp * Sp) * 2);        // FY
    plan(B8((long)Sp * Fp) * 2);        // IY
    plan(B8((long)Hp * Sp) * 2);        // IX
    const int npsf = same ? n1 : n1 + n2;
    plan(B8((long)npsf * Sp * Sp));     // padded PSFs
    plan(B8((long)npsf * Sp * Hp) * 2); // Y1 (x-transformed)
    plan(B8((long)npsf * plane) * 2);   // spectra
    plan(B8((long)npairs * plane) * 2); // Z
    plan(B8((long)npairs * Sp * Hp) * 2); // U
    plan(B8((long)npairs * Sp * Sp));   // windows
    plan((size_t)npairs * 8);
    IMCOM_TRY(ws_reserve(ctx, total + 8192));
    double *FXc = (double *)ws_take(ctx, B8((long)Sp * Hp) * 2), *FXs = FXc + (long)Sp * Hp;
    double *FYc = (double *)ws_take(ctx, B8((long)Fp * Sp) * 2), *FYs = FYc + (long)Fp * Sp;
    double *IYc = (double *)ws_take(ctx, B8((long)Sp * Fp) * 2), *IYs = IYc + (long)Sp * Fp;
    double *IXc = (double *)ws_take(ctx, B8((long)Hp * Sp) * 2), *IXs = IXc + (long)Hp * Sp;
    double *X = (double *)ws_take(ctx, B8((long)npsf * Sp * Sp));
    double *Y1 = (double *)ws_take(ctx, B8((long)npsf * Sp * Hp) * 2);
    double *R = (double *)ws_take(ctx, B8((long)npsf * plane) * 2);
    double *Z = (double *)ws_take(ctx, B8((long)npairs * plane) * 2);
    double *U = (double *)ws_take(ctx, B8((long)npairs * Sp * Hp) * 2);
    double *W = (double *)ws_take(ctx, B8((long)npairs * Sp * Sp));
    int *pairs_dev = (int *)ws_take(ctx, (size_t)npairs * 8);
    if (!FXc || !FYc || !IYc || !IXc || !X || !Y1 || !R || !Z || !U || !W || !pairs_dev) {
        set_error("internal: workspace plan too small");
        return IMCOM_ERR_NOMEM;
    }
    hipStream_t st = ctx->stream;
    ProfScope ps(ctx, "psf_overlap");
    // second group's PSFs are stored after the first's; remap q
    std::vector<int> pr(2 * (size_t)npairs);
    for (int t = 0; t < npairs; t++) { pr[2 * t] = pairs_host[2 * t]; pr[2 * t + 1] = pairs_host[2 * t + 1] + (same ? 0 : n1); }
    IMCOM_HIP_CHECK(hipMemcpyAsync(pairs_dev, pr.data(), pr.size() * 4, hipMemcpyHostToDevice, st));
    IMCOM_HIP_CHECK(hipStreamSynchronize(st));  // pr is a local
    auto dft = [&](int kind, int rows, int cols, int rv, int cv, double *c, double *s) {
        hipLaunchKernelGGL(dft_matrix_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, st, kind, rows, cols, rv, cv, nfft, nc, c, s);
    };
    dft(0, Sp, Hp, nsamp, nh, FXc, FXs);
    dft(1, Fp, Sp, nfft, nsamp, FYc, FYs);
    dft(2, Sp, Fp, nsamp, nfft, IYc, IYs);
    dft(3, Hp, Sp, nh, nsamp, IXc, IXs);
    hipLaunchKernelGGL(pad_psf_kernel, dim3((Sp + 255) / 256, Sp, n1), dim3(256), 0, st, psf1, nsamp, X, Sp);
    if (!same) hipLaunchKernelGGL(pad_psf_kernel, dim3((Sp + 255) / 256, Sp, n2), dim3(256), 0, st, psf2, nsamp, X + (long)n1 * Sp * Sp, Sp);
    IMCOM_TRY(check_launch("psf_overlap prologue"));
    const long sX = (long)Sp * Sp, sY1 = 2L * Sp * Hp, sR = 2L * plane;
    // forward along x: Y1 = X (FXc + i FXs)              [Sp x Sp] . [Sp x Hp]
    IMCOM_TRY(launch_gemm(ctx, false, true, Sp, Hp, Sp, npsf, X, Sp, sX, FXc, Hp, 0, Y1, Hp, sY1, 1.0, 0.0));
    IMCOM_TRY(launch_gemm(ctx, false, true, Sp, Hp, Sp, npsf, X, Sp, sX, FXs, Hp, 0, Y1 + (long)Sp * Hp, Hp, sY1, 1.0, 0.0));
    // forward along y: R = (FYc + i FYs) Y1               [Fp x Sp] . [Sp x Hp]
    double *Y1r = Y1, *Y1i = Y1 + (long)Sp * Hp, *Rr = R, *Ri = R + plane;
    IMCOM_TRY(launch_gemm(ctx, false, true, Fp, Hp, Sp, npsf, FYc, Sp, 0, Y1r, Hp, sY1, Rr, Hp, sR, 1.0, 0.0));
    IMCOM_TRY(launch_gemm(ctx, false, true, Fp, Hp, Sp, npsf, FYs, Sp, 0, Y1i, Hp, sY1, Rr, Hp, sR, -1.0, 1.0));
    IMCOM_TRY(launch_gemm(ctx, false, true, Fp, Hp, Sp, npsf, FYc, Sp, 0, Y1i, Hp, sY1, Ri, Hp, sR, 1.0, 0.0));
    IMCOM_TRY(launch_gemm(ctx, false, true, Fp, Hp, Sp, npsf, FYs, Sp, 0, Y1r, Hp, sY1, Ri, Hp, sR, 1.0, 1.0));
    // spectra product
    hipLaunchKernelGGL(cmul_conj_kernel, dim3((unsigned)((plane + 255) / 256), npairs), dim3(256), 0, st, R, R, pairs_dev, plane, Hp,
                       nfft, amp_penalty ? amp_penalty[0] : 0.0, amp_penalty ? amp_penalty[1] : 0.0, Z);
    IMCOM_TRY(check_launch("cmul_conj_kernel"));
    // inverse along y on the kept rows: U = (IYc + i IYs) Z    [Sp x Fp] . [Fp x Hp]
    const long sZ = 2L * plane, sU = 2L * Sp * Hp;
    double *Zr = Z, *Zi = Z + plane, *Ur = U, *Ui = U + (long)Sp * Hp;
    IMCOM_TRY(launch_gemm(ctx, false, true, Sp, Hp, Fp, npairs, IYc, Fp, 0, Zr, Hp, sZ, Ur, Hp, sU, 1.0, 0.0));
    IMCOM_TRY(launch_gemm(ctx, false, true, Sp, Hp, Fp, npairs, IYs, Fp, 0, Zi, Hp, sZ, Ur, Hp, sU, -1.0, 1.0));
    IMCOM_TRY(launch_gemm(ctx, false, true, Sp, Hp, Fp, npairs, IYc, Fp, 0, Zi, Hp, sZ, Ui, Hp, sU, 1.0, 0.0));
    IMCOM_TRY(launch_gemm(ctx, false, true, Sp, Hp, Fp, npairs, IYs, Fp, 0, Zr, Hp, sZ, Ui, Hp, sU, 1.0, 1.0));
    // inverse along x on the kept columns (Hermitian half-spectrum weights folded into IX)
    IMCOM_TRY(launch_gemm(ctx, false, true, Sp, Sp, Hp, npairs, Ur, Hp, sU, IXc, Sp, 0, W, Sp, (long)Sp * Sp, 1.0, 0.0));
    IMCOM_TRY(launch_gemm(ctx, false, true, Sp, Sp, Hp, npairs, Ui, Hp, sU, IXs, Sp, 0, W, Sp, (long)Sp * Sp, 1.0, 1.0));
    const int ng = nsamp + 12;
    hipLaunchKernelGGL(crop_table_kernel, dim3((ng + 255) / 256, ng, npairs), dim3(256), 0, st, W, Sp, nsamp, tables);
    return check_launch("crop_table_kernel");
}

extern "C" long imcom_psf_spectra_size(int nsamp, int nfft)
{
    FftPlan pl;
    if (nsamp < 1 || nfft < 2 * nsamp || nfft % 2 || fft_force_gemm() || !fft_plan(nfft, &pl)) return 0;
    return 2L * (nfft / 2 + 1) * nfft;
}

extern "C" int imcom_psf_spectra(imcom_ctx *ctx, const double *psf, int n, int nsamp, int nfft, double *spectra)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(psf && spectra && n >= 1, "null pointer / empty set");
    IMCOM_REQUIRE(imcom_psf_spectra_size(nsamp, nfft) > 0, "no butterfly plan for nfft=%d (use imcom_psf_overlap)", nfft);
    FftPlan pl;
    fft_plan(nfft, &pl);
    IMCOM_TRY(ws_reserve(ctx, fft_forward_ws(n, nsamp, nfft) + 8192));
    ProfScope ps(ctx, "psf_spectra");
    return fft_forward(ctx, pl, psf, n, nsamp, (cplx *)spectra);
}

extern "C" int imcom_psf_overlap_spectra_slots(imcom_ctx *ctx, const double *spec1, int n1, const double *spec2, int n2, int nsamp,
                                               int nfft, const int *pairs_host, int npairs, const double *amp_penalty,
                                               const int *win_host, const int *slots_host, int nslots, double *tables)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(spec1 && spec2 && pairs_host && tables && npairs >= 1, "null pointer / no pairs");
    IMCOM_REQUIRE(imcom_psf_spectra_size(nsamp, nfft) > 0, "no butterfly plan for nfft=%d (use imcom_psf_overlap)", nfft);
    for (int t = 0; t < npairs; t++)
        IMCOM_REQUIRE(pairs_host[2 * t] >= 0 && pairs_host[2 * t] < n1 && pairs_host[2 * t + 1] >= 0 && pairs_host[2 * t + 1] < n2,
                      "pair %d out of range", t);
    if (win_host)
        for (int t = 0; t < npairs; t++) {
            const int *w = win_host + 4 * (size_t)t;
            IMCOM_REQUIRE(0 <= w[0] && w[0] < w[1] && w[1] <= nsamp && 0 <= w[2] && w[2] < w[3] && w[3] <= nsamp, "window %d out of range", t);
        }
    if (slots_host)
        for (int t = 0; t < npairs; t++) IMCOM_REQUIRE(slots_host[t] >= 0 && slots_host[t] < nslots, "slot %d of pair %d outside the arena of %d tables", slots_host[t], t, nslots);
    FftPlan pl;
    fft_plan(nfft, &pl);
    // in chunks of pairs, so that the intermediate (nsamp x nh complex per pair) stays within ~4 GB however many
    // tables a caller asks for at once; the chunks run back to back on the stream and reuse the workspace in order
    const size_t per_pair = (size_t)(nsamp + 1) * v_stride(nfft) * 16;
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)npairs, ((size_t)4 << 30) / per_pair));
    if (const char *e = getenv("IMCOM_FFT_CHUNK_PAIRS")) chunk = std::max(1, std::min(npairs, atoi(e)));  // tuning runs
    IMCOM_TRY(ws_reserve(ctx, fft_inverse_ws(chunk, nsamp, nfft) + 8192));
    ProfScope ps(ctx, "psf_overlap");
    const size_t tab = (size_t)(nsamp + 12) * (nsamp + 12);
    for (int p0 = 0; p0 < npairs; p0 += chunk) {
        ctx->ws_used = 0;
        IMCOM_TRY(fft_inverse(ctx, pl, (const cplx *)spec1, (const cplx *)spec2, pairs_host + 2 * (size_t)p0, std::min(chunk, npairs - p0), nsamp,
                              amp_penalty, slots_host ? tables : tables + (size_t)p0 * tab, win_host ? win_host + 4 * (size_t)p0 : nullptr,
                              slots_host ? slots_host + p0 : nullptr));
    }
    return IMCOM_OK;
}

extern "C" int imcom_psf_overlap_spectra_win(imcom_ctx *ctx, const double *spec1, int n1, const double *spec2, int n2, int nsamp,
                                             int nfft, const int *pairs_host, int npairs, const double *amp_penalty,
                                             const int *win_host, double *tables)
{
    return imcom_psf_overlap_spectra_slots(ctx, spec1, n1, spec2, n2, nsamp, nfft, pairs_host, npairs, amp_penalty, win_host, nullptr, 0, tables);
}

extern "C" int imcom_psf_overlap_spectra(imcom_ctx *ctx, const double *spec1, int n1, const double *spec2, int n2, int nsamp,
                                         int nfft, const int *pairs_host, int npairs, const double *amp_penalty, double *tables)
{
    return imcom_psf_overlap_spectra_win(ctx, spec1, n1, spec2, n2, nsamp, nfft, pairs_host, npairs, amp_penalty, nullptr, tables);
}
