// psf_overlap.hip -- PSF cross-correlation tables on the device.
//
// Replaces PSFGrp.accel_pad_and_rfft2 (reference src/pyimcom/psfutil.py:943-986) and
// PSFOvl._build_psfovl / accel_irfft2_and_extract (1244-1294, 1178-1242):
//     table[p,q] = roll(irfft2(rfft2(pad(psf1[p])) * conj(rfft2(pad(psf2[q])))), nc)[:nsamp, :nsamp]
// written with the 6-pixel zero border the interpolators expect.
//
// Two formulations.  (1) Butterfly path (default whenever nfft = product of 4, 2, 3, 5 and <= 1024): mixed-radix
// Stockham FFTs in LDS, see below -- 7 us per cfg-2 table (nfft 768).
// (2) Dense-DFT path (general fallback, IMCOM_PSF_OVERLAP=gemm): the zero-padded 2-D DFTs as dense real matrix
// products with exact twiddle matrices (integer argument reduction mod nfft, then cos/sin of a multiple of
// pi/nfft) on the fp64 MFMA tile engine of gemm_f64.hip; only the kept nsamp x nsamp window of the inverse is
// computed; O(nfft * nsamp * nh) per stage, 51 us per cfg-2 table.  Both agree with numpy's FFTs to ~1e-15.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"
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
// Butterfly path: mixed-radix Stockham FFTs in LDS (radices 4, 2, 3, 5), used whenever nfft factors into them and
// the lines fit (nfft <= 1024); the dense-DFT GEMM form above stays as the general fallback and as an independent
// cross-check (IMCOM_PSF_OVERLAP=gemm).  A workgroup transforms L (8 or 4) lines at once, all in one LDS buffer: in
// every stage each thread first reads all of its butterflies into registers, then -- after a barrier -- writes
// them back in Stockham order, so no second buffer and no digit reversal are needed.
//   forward:  rows two-for-one (two real PSF rows ride as one complex line), then columns on tiles of L
//   inverse:  columns of R1 conj(R2) (the product is formed on load), kept rows only; then rows two-for-one from
//             the Hermitian half back to two real window rows, rolled by nc and cropped on store.
// lines per workgroup: template parameter L of the kernels, 8 where the plan allows (the per-workgroup costs -- twiddle
// table, barriers -- are shared by more lines: 0.39 against 0.43 ms for 36 cfg-2 tables), else 4
constexpr int FFT_MAXIT = 8;   // butterflies per thread and stage (nfft * L / (radix * 256) <= 8)
constexpr int FFT_MAXN = 1024;
constexpr int FFT_NT = 1024;    // threads per workgroup of the line kernels

struct FftPlan { int n, nst, radix[12], lines; };

typedef double2 cplx;
__device__ __forceinline__ cplx cmulf(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

template <int R, bool INV> __device__ __forceinline__ void small_dft(cplx (&v)[R]);
template <> __device__ __forceinline__ void small_dft<2, false>(cplx (&v)[2])
{ const cplx a = v[0], b = v[1]; v[0] = make_double2(a.x + b.x, a.y + b.y); v[1] = make_double2(a.x - b.x, a.y - b.y); }
template <> __device__ __forceinline__ void small_dft<2, true>(cplx (&v)[2]) { small_dft<2, false>(v); }
template <bool INV> __device__ __forceinline__ void dft4(cplx (&v)[4])
{
    const cplx t0 = make_double2(v[0].x + v[2].x, v[0].y + v[2].y), t1 = make_double2(v[0].x - v[2].x, v[0].y - v[2].y);
    const cplx t2 = make_double2(v[1].x + v[3].x, v[1].y + v[3].y), d = make_double2(v[1].x - v[3].x, v[1].y - v[3].y);
    const cplx t3 = INV ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);  // d * (+i) or d * (-i)
    v[0] = make_double2(t0.x + t2.x, t0.y + t2.y);
    v[2] = make_double2(t0.x - t2.x, t0.y - t2.y);
    v[1] = make_double2(t1.x + t3.x, t1.y + t3.y);
    v[3] = make_double2(t1.x - t3.x, t1.y - t3.y);
}
template <> __device__ __forceinline__ void small_dft<4, false>(cplx (&v)[4]) { dft4<false>(v); }
template <> __device__ __forceinline__ void small_dft<4, true>(cplx (&v)[4]) { dft4<true>(v); }
template <bool INV> __device__ __forceinline__ void dft3(cplx (&v)[3])
{
    const double s3 = INV ? 0.8660254037844386 : -0.8660254037844386;  // sin(-+ 2 pi / 3)
    const cplx t = make_double2(v[1].x + v[2].x, v[1].y + v[2].y), u = make_double2(v[1].x - v[2].x, v[1].y - v[2].y);
    const cplx m = make_double2(v[0].x - 0.5 * t.x, v[0].y - 0.5 * t.y), iu = make_double2(-s3 * u.y, s3 * u.x);  // i s3 u
    v[0] = make_double2(v[0].x + t.x, v[0].y + t.y);
    v[1] = make_double2(m.x + iu.x, m.y + iu.y);
    v[2] = make_double2(m.x - iu.x, m.y - iu.y);
}
template <> __device__ __forceinline__ void small_dft<3, false>(cplx (&v)[3]) { dft3<false>(v); }
template <> __device__ __forceinline__ void small_dft<3, true>(cplx (&v)[3]) { dft3<true>(v); }
template <bool INV> __device__ __forceinline__ void dft5(cplx (&v)[5])
{
    // y_u = sum_t v_t w^(u t), w = exp(-+ 2 pi i / 5)
    const double c1 = 0.30901699437494745, c2 = -0.8090169943749475;
    const double s1 = INV ? 0.9510565162951535 : -0.9510565162951535, s2 = INV ? 0.5877852522924731 : -0.5877852522924731;
    const cplx a = make_double2(v[1].x + v[4].x, v[1].y + v[4].y), b = make_double2(v[1].x - v[4].x, v[1].y - v[4].y);
    const cplx c = make_double2(v[2].x + v[3].x, v[2].y + v[3].y), d = make_double2(v[2].x - v[3].x, v[2].y - v[3].y);
    const cplx m1 = make_double2(v[0].x + c1 * a.x + c2 * c.x, v[0].y + c1 * a.y + c2 * c.y);
    const cplx m2 = make_double2(v[0].x + c2 * a.x + c1 * c.x, v[0].y + c2 * a.y + c1 * c.y);
    const cplx n1 = make_double2(-(s1 * b.y + s2 * d.y), s1 * b.x + s2 * d.x);  // i (s1 b + s2 d)
    const cplx n2 = make_double2(-(s2 * b.y - s1 * d.y), s2 * b.x - s1 * d.x);  // i (s2 b - s1 d)
    v[0] = make_double2(v[0].x + a.x + c.x, v[0].y + a.y + c.y);
    v[1] = make_double2(m1.x + n1.x, m1.y + n1.y);
    v[4] = make_double2(m1.x - n1.x, m1.y - n1.y);
    v[2] = make_double2(m2.x + n2.x, m2.y + n2.y);
    v[3] = make_double2(m2.x - n2.x, m2.y - n2.y);
}
template <> __device__ __forceinline__ void small_dft<5, false>(cplx (&v)[5]) { dft5<false>(v); }
template <> __device__ __forceinline__ void small_dft<5, true>(cplx (&v)[5]) { dft5<true>(v); }

// one Stockham stage of radix R on L lines of length n in `buf` ([line][n]); Ns = product of the earlier radices;
// tw[k] = exp(-2 pi i k / n)
template <int R, bool INV, int L>
__device__ __forceinline__ void fft_stage(cplx *buf, int n, int Ns, const cplx *tw)
{
    // butterfly j of a line is handled by thread j mod 256, line after line: no index divisions (fft_plan guarantees
    // n / R <= 256 * FFT_MAXIT / L); Ns is a power of two until the first radix-3 / 5 stage
    // the workgroup is FFT_NT / 256 groups of 256 threads, each taking its share of the lines (sixteen waves per CU hide
    // the LDS latency better than four with four times the butterflies each: 0.31 -> 0.25 ms for 36 cfg-2 tables)
    constexpr int Q = FFT_MAXIT / L, LH = L / (FFT_NT / 256);
    const int nb = n / R, step = n / (Ns * R), tid = threadIdx.x & 255, l0 = (threadIdx.x >> 8) * LH;
    const bool pow2 = (Ns & (Ns - 1)) == 0;
    cplx v[LH * Q][R];
#pragma unroll
    for (int lh = 0; lh < LH; lh++)
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int j = tid + q * 256, line = l0 + lh;
            if (j < nb) {
                const int k = pow2 ? (j & (Ns - 1)) : j % Ns;
                const cplx *x = buf + line * n;
#pragma unroll
                for (int t = 0; t < R; t++) {
                    cplx a = x[j + t * nb];
                    if (t > 0) {
                        cplx w = tw[t * k * step];
                        if (INV) w.y = -w.y;
                        a = cmulf(a, w);
                    }
                    v[lh * Q + q][t] = a;
                }
                small_dft<R, INV>(v[lh * Q + q]);
            }
        }
    __syncthreads();
#pragma unroll
    for (int lh = 0; lh < LH; lh++)
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int j = tid + q * 256, line = l0 + lh;
            if (j < nb) {
                const int k = pow2 ? (j & (Ns - 1)) : j % Ns;
                cplx *x = buf + line * n + (j - k) * R + k;
#pragma unroll
                for (int u = 0; u < R; u++) x[u * Ns] = v[lh * Q + q][u];
            }
        }
    __syncthreads();
}

template <bool INV, int L>
__device__ __forceinline__ void fft_lines(cplx *buf, const FftPlan &pl, const cplx *tw)
{
    int Ns = 1;
    for (int st = 0; st < pl.nst; st++) {
        const int r = pl.radix[st];
        if (r == 4) fft_stage<4, INV, L>(buf, pl.n, Ns, tw);
        else if (r == 2) fft_stage<2, INV, L>(buf, pl.n, Ns, tw);
        else if (r == 3) fft_stage<3, INV, L>(buf, pl.n, Ns, tw);
        else fft_stage<5, INV, L>(buf, pl.n, Ns, tw);
        Ns *= r;
    }
}

__global__ void fft_twiddle_kernel(int n, cplx *__restrict__ tw)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double c, s;
    twiddle(k, n, &c, &s);
    tw[k] = make_double2(c, -s);
}

// forward, along x: rows 2l, 2l+1 of PSF p as one complex line; Y1[p][kx][row], kx < nh
template <int L>
__global__ __launch_bounds__(FFT_NT) void fft_fwd_rows_kernel(const double *__restrict__ psf, int ns, FftPlan pl,
                                                           const cplx *__restrict__ tw, cplx *__restrict__ Y1)
{
    extern __shared__ cplx fbuf[];
    const int n = pl.n, nh = n / 2 + 1, p = blockIdx.y;
    cplx *twl = fbuf + L * n;  // the twiddle table rides in LDS behind the lines
    for (int e = threadIdx.x; e < n; e += FFT_NT) twl[e] = tw[e];
    const double *img = psf + (long)p * ns * ns;
#pragma unroll
    for (int line = 0; line < L; line++)
#pragma unroll 3
        for (int x = threadIdx.x; x < n; x += FFT_NT) {
        const int e = line * n + x, r0 = 2 * (blockIdx.x * L + line);
        double re = 0.0, im = 0.0;
        if (x < ns) {
            if (r0 < ns) re = img[(long)r0 * ns + x];
            if (r0 + 1 < ns) im = img[(long)(r0 + 1) * ns + x];
        }
        fbuf[e] = make_double2(re, im);
    }
    __syncthreads();
    fft_lines<false, L>(fbuf, pl, twl);
    for (int e = threadIdx.x; e < L * nh; e += FFT_NT) {  // Y1 is stored [kx][row]: the 2 L rows of this block are contiguous
        const int k = e / L, line = e - k * L, r0 = 2 * (blockIdx.x * L + line);
        if (r0 >= ns) continue;
        const cplx zk = fbuf[line * n + k], zm = fbuf[line * n + (k ? n - k : 0)];
        Y1[((long)p * nh + k) * ns + r0] = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        if (r0 + 1 < ns) Y1[((long)p * nh + k) * ns + r0 + 1] = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
    }
}

// forward, along y: L columns of Y1[p] (rows >= ns are zero) -> R[p][kx][ky]
template <int L>
__global__ __launch_bounds__(FFT_NT) void fft_fwd_cols_kernel(const cplx *__restrict__ Y1, int ns, FftPlan pl,
                                                           const cplx *__restrict__ tw, cplx *__restrict__ R)
{
    extern __shared__ cplx fbuf[];
    const int n = pl.n, nh = n / 2 + 1, p = blockIdx.y, kx0 = blockIdx.x * L;
    cplx *twl = fbuf + L * n;  // the twiddle table rides in LDS behind the lines
    for (int e = threadIdx.x; e < n; e += FFT_NT) twl[e] = tw[e];
#pragma unroll
    for (int c = 0; c < L; c++)
#pragma unroll 3
        for (int y = threadIdx.x; y < n; y += FFT_NT) {
        const int e = c * n + y;
        cplx v = make_double2(0.0, 0.0);
        if (y < ns && kx0 + c < nh) v = Y1[((long)p * nh + kx0 + c) * ns + y];
        fbuf[e] = v;
    }
    __syncthreads();
    fft_lines<false, L>(fbuf, pl, twl);
#pragma unroll
    for (int c = 0; c < L; c++)
#pragma unroll 3
        for (int ky = threadIdx.x; ky < n; ky += FFT_NT) {  // spectra are stored [kx][ky]: whole lines
        const int e = c * n + ky;
        if (kx0 + c < nh) R[((long)p * nh + kx0 + c) * n + ky] = fbuf[e];
    }
}

// inverse, along y: L columns of R1[p] conj(R2[q]) (x the squared Fourier-mode weight) -> V[t][kx][y'] for the
// kept rows y' < ns (source row (y' - nc) mod n: the roll of psfutil.py:1225-1232)
template <int L>
__global__ __launch_bounds__(FFT_NT) void fft_inv_cols_kernel(const cplx *__restrict__ Ra, const cplx *__restrict__ Rb,
                                                           const int *__restrict__ pairs, int ns,
                                                           FftPlan pl, const cplx *__restrict__ tw, double amp0, double amps,
                                                           cplx *__restrict__ V)
{
    extern __shared__ cplx fbuf[];
    const int n = pl.n, nh = n / 2 + 1, t = blockIdx.y, kx0 = blockIdx.x * L, nc = ns / 2;
    cplx *twl = fbuf + L * n;  // the twiddle table rides in LDS behind the lines
    for (int e = threadIdx.x; e < n; e += FFT_NT) twl[e] = tw[e];
    const cplx *R1 = Ra + (long)pairs[2 * t] * n * nh, *R2 = Rb + (long)pairs[2 * t + 1] * n * nh;
#pragma unroll
    for (int c = 0; c < L; c++)
#pragma unroll 3
        for (int ky = threadIdx.x; ky < n; ky += FFT_NT) {
        const int e = c * n + ky, kx = kx0 + c;
        cplx z = make_double2(0.0, 0.0);
        if (kx < nh) {
            const cplx a = R1[(long)kx * n + ky], b = R2[(long)kx * n + ky];
            double w2 = 1.0;
            if (amp0 != 0.0) {
                double uy = (double)ky / (double)n, ux = (double)kx / (double)n;
                if (uy > 0.5) uy -= 1.0;
                if (ux > 0.5) ux -= 1.0;
                const double w = 1.0 + amp0 * exp(-2.0 * M_PI * M_PI * (ux * ux + uy * uy) * (amps * amps));
                w2 = w * w;
            }
            z = make_double2((a.x * b.x + a.y * b.y) * w2, (a.y * b.x - a.x * b.y) * w2);
        }
        fbuf[e] = z;
    }
    __syncthreads();
    fft_lines<true, L>(fbuf, pl, twl);
#pragma unroll
    for (int c = 0; c < L; c++)
#pragma unroll 3
        for (int yp = threadIdx.x; yp < ns; yp += FFT_NT) {  // V is stored [kx][y']
        if (kx0 + c < nh) V[((long)t * nh + kx0 + c) * ns + yp] = fbuf[c * n + yp - nc + (yp < nc ? n : 0)];
    }
}

// inverse, along x: window rows 2l, 2l+1 of pair t from their Hermitian halves as one complex line; the real and the
// imaginary part of the result are the two rows.  numpy's c2r ignores the imaginary parts of the DC and Nyquist
// bins; so does this.  Stored rolled by nc, cropped to ns, scaled by 1/n^2, inside the 6-sample zero border.
template <int L>
__global__ __launch_bounds__(FFT_NT) void fft_inv_rows_kernel(const cplx *__restrict__ V, int ns, FftPlan pl,
                                                           const cplx *__restrict__ tw, double *__restrict__ tables)
{
    extern __shared__ cplx fbuf[];
    const int n = pl.n, nh = n / 2 + 1, t = blockIdx.y, nc = ns / 2, ng = ns + 12;
    cplx *twl = fbuf + L * n;  // the twiddle table rides in LDS behind the lines
    for (int e = threadIdx.x; e < n; e += FFT_NT) twl[e] = tw[e];
#pragma unroll 6
    for (int e = threadIdx.x; e < L * n; e += FFT_NT) {  // the 2 L rows of this block are contiguous in V[kx][y']
        const int k = e / L, line = e - k * L, r0 = 2 * (blockIdx.x * L + line);
        const int kk = k < nh ? k : n - k;
        const bool edge = kk == 0 || 2 * kk == n;
        cplx a = make_double2(0.0, 0.0), b = a;
        if (r0 < ns) a = V[((long)t * nh + kk) * ns + r0];
        if (r0 + 1 < ns) b = V[((long)t * nh + kk) * ns + r0 + 1];
        if (k >= nh) { a.y = -a.y; b.y = -b.y; }
        if (edge) { a.y = 0.0; b.y = 0.0; }
        fbuf[line * n + k] = make_double2(a.x - b.y, a.y + b.x);
    }
    __syncthreads();
    fft_lines<true, L>(fbuf, pl, twl);
    const double scale = 1.0 / ((double)n * (double)n);
#pragma unroll
    for (int line = 0; line < L; line++)
#pragma unroll 3
        for (int xp = threadIdx.x; xp < ns; xp += FFT_NT) {
        const int r0 = 2 * (blockIdx.x * L + line);
        if (r0 >= ns) continue;
        const cplx z = fbuf[line * n + xp - nc + (xp < nc ? n : 0)];
        tables[((long)t * ng + 6 + r0) * ng + 6 + xp] = z.x * scale;
        if (r0 + 1 < ns) tables[((long)t * ng + 6 + r0 + 1) * ng + 6 + xp] = z.y * scale;
    }
}

// nfft = product of radices 4, 2, 3, 5 with lines that fit the butterfly kernels?
static bool fft_plan(int n, FftPlan *pl)
{
    if (n > FFT_MAXN) return false;
    pl->n = n;
    pl->nst = 0;
    int r = n;
    const int cand[4] = {4, 2, 3, 5};
    for (int ci = 0; ci < 4; ci++)
        while (r % cand[ci] == 0 && !(cand[ci] == 4 && r % 4 != 0)) {
            if (pl->nst >= 12) return false;
            pl->radix[pl->nst++] = cand[ci];
            r /= cand[ci];
        }
    if (r != 1) return false;
    for (int lines = 8; lines >= 4; lines /= 2) {
        bool ok = (size_t)(lines + 1) * n * 16 <= 160 * 1024;
        for (int st = 0; st < pl->nst; st++) ok = ok && n / pl->radix[st] <= 256 * (FFT_MAXIT / lines);
        if (ok) { pl->lines = lines; return true; }
    }
    return false;
}


static bool fft_force_gemm()
{
    static const bool f = getenv("IMCOM_PSF_OVERLAP") && !strcmp(getenv("IMCOM_PSF_OVERLAP"), "gemm");
    return f;
}

static size_t fft_lds_bytes(const FftPlan &pl) { return (size_t)(pl.lines + 1) * pl.n * 16; }  // lines + twiddle table

template <int L>
static int fft_set_lds_l(size_t lds)
{
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)fft_fwd_rows_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)fft_fwd_cols_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)fft_inv_cols_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    IMCOM_HIP_CHECK(hipFuncSetAttribute((const void *)fft_inv_rows_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return IMCOM_OK;
}

static int fft_set_lds(const FftPlan &pl)
{
    const size_t lds = fft_lds_bytes(pl);
    if (lds <= 48 * 1024) return IMCOM_OK;
    return pl.lines == 8 ? fft_set_lds_l<8>(lds) : fft_set_lds_l<4>(lds);
}

static size_t fft_forward_ws(int n, int nsamp, int nfft) { return (size_t)nfft * 16 + (size_t)n * nsamp * (nfft / 2 + 1) * 16 + 1024; }
static size_t fft_inverse_ws(int npairs, int nsamp, int nfft) { return (size_t)nfft * 16 + (size_t)npairs * nsamp * (nfft / 2 + 1) * 16 + (size_t)npairs * 8 + 1024; }

// spectra R[n][nh][nfft] (complex) of n sampled PSFs; the caller has reserved fft_forward_ws() of workspace
static int fft_forward(imcom_ctx *ctx, const FftPlan &pl, const double *psf, int n, int nsamp, cplx *R)
{
    const int nfft = pl.n, nh = nfft / 2 + 1;
    cplx *tw = (cplx *)ws_take(ctx, (size_t)nfft * 16);
    cplx *Y1 = (cplx *)ws_take(ctx, (size_t)n * nsamp * nh * 16);
    if (!tw || !Y1) { set_error("internal: workspace plan too small"); return IMCOM_ERR_NOMEM; }
    IMCOM_TRY(fft_set_lds(pl));
    hipStream_t st = ctx->stream;
    const size_t lds = fft_lds_bytes(pl);
    const int FL = pl.lines, row_blocks = ((nsamp + 1) / 2 + FL - 1) / FL, col_blocks = (nh + FL - 1) / FL;
    hipLaunchKernelGGL(fft_twiddle_kernel, dim3((nfft + 255) / 256), dim3(256), 0, st, nfft, tw);
    if (FL == 8) {
        hipLaunchKernelGGL(fft_fwd_rows_kernel<8>, dim3(row_blocks, n), dim3(FFT_NT), lds, st, psf, nsamp, pl, tw, Y1);
        hipLaunchKernelGGL(fft_fwd_cols_kernel<8>, dim3(col_blocks, n), dim3(FFT_NT), lds, st, Y1, nsamp, pl, tw, R);
    } else {
        hipLaunchKernelGGL(fft_fwd_rows_kernel<4>, dim3(row_blocks, n), dim3(FFT_NT), lds, st, psf, nsamp, pl, tw, Y1);
        hipLaunchKernelGGL(fft_fwd_cols_kernel<4>, dim3(col_blocks, n), dim3(FFT_NT), lds, st, Y1, nsamp, pl, tw, R);
    }
    return check_launch("psf spectra (butterfly path)");
}

// tables[t] from spectra pairs (Ra[pairs[2t]], Rb[pairs[2t+1]]); the caller has reserved fft_inverse_ws()
static int fft_inverse(imcom_ctx *ctx, const FftPlan &pl, const cplx *Ra, const cplx *Rb, const int *pairs_host, int npairs,
                       int nsamp, const double *amp_penalty, double *tables)
{
    const int nfft = pl.n, nh = nfft / 2 + 1, ng = nsamp + 12;
    cplx *tw = (cplx *)ws_take(ctx, (size_t)nfft * 16);
    cplx *V = (cplx *)ws_take(ctx, (size_t)npairs * nsamp * nh * 16);
    int *pairs_dev = (int *)ws_take(ctx, (size_t)npairs * 8);
    if (!tw || !V || !pairs_dev) { set_error("internal: workspace plan too small"); return IMCOM_ERR_NOMEM; }
    IMCOM_TRY(fft_set_lds(pl));
    hipStream_t st = ctx->stream;
    IMCOM_TRY(upload(ctx, pairs_dev, pairs_host, 2 * (size_t)npairs));  // through the pinned ring: no stream drain
    const size_t lds = fft_lds_bytes(pl);
    const int FL = pl.lines, row_blocks = ((nsamp + 1) / 2 + FL - 1) / FL, col_blocks = (nh + FL - 1) / FL;
    const double a0 = amp_penalty ? amp_penalty[0] : 0.0, a1 = amp_penalty ? amp_penalty[1] : 0.0;
    hipLaunchKernelGGL(fft_twiddle_kernel, dim3((nfft + 255) / 256), dim3(256), 0, st, nfft, tw);
    IMCOM_HIP_CHECK(hipMemsetAsync(tables, 0, (size_t)npairs * ng * ng * 8, st));
    if (FL == 8) {
        hipLaunchKernelGGL(fft_inv_cols_kernel<8>, dim3(col_blocks, npairs), dim3(FFT_NT), lds, st, Ra, Rb, pairs_dev, nsamp, pl, tw, a0, a1, V);
        hipLaunchKernelGGL(fft_inv_rows_kernel<8>, dim3(row_blocks, npairs), dim3(FFT_NT), lds, st, V, nsamp, pl, tw, tables);
    } else {
        hipLaunchKernelGGL(fft_inv_cols_kernel<4>, dim3(col_blocks, npairs), dim3(FFT_NT), lds, st, Ra, Rb, pairs_dev, nsamp, pl, tw, a0, a1, V);
        hipLaunchKernelGGL(fft_inv_rows_kernel<4>, dim3(row_blocks, npairs), dim3(FFT_NT), lds, st, V, nsamp, pl, tw, tables);
    }
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

extern "C" int imcom_psf_overlap_spectra(imcom_ctx *ctx, const double *spec1, int n1, const double *spec2, int n2, int nsamp,
                                         int nfft, const int *pairs_host, int npairs, const double *amp_penalty, double *tables)
{
    if (!ctx) { set_error("null context"); return IMCOM_ERR_ARG; }
    IMCOM_HIP_CHECK(hipSetDevice(ctx->device));
    IMCOM_REQUIRE(spec1 && spec2 && pairs_host && tables && npairs >= 1, "null pointer / no pairs");
    IMCOM_REQUIRE(imcom_psf_spectra_size(nsamp, nfft) > 0, "no butterfly plan for nfft=%d (use imcom_psf_overlap)", nfft);
    for (int t = 0; t < npairs; t++)
        IMCOM_REQUIRE(pairs_host[2 * t] >= 0 && pairs_host[2 * t] < n1 && pairs_host[2 * t + 1] >= 0 && pairs_host[2 * t + 1] < n2,
                      "pair %d out of range", t);
    FftPlan pl;
    fft_plan(nfft, &pl);
    // in chunks of pairs, so that the intermediate (nsamp x nh complex per pair) stays within ~4 GB however many
    // tables a caller asks for at once; the chunks run back to back on the stream and reuse the workspace in order
    const size_t per_pair = (size_t)nsamp * (nfft / 2 + 1) * 16;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)npairs, ((size_t)4 << 30) / per_pair));
    IMCOM_TRY(ws_reserve(ctx, fft_inverse_ws(chunk, nsamp, nfft) + 8192));
    ProfScope ps(ctx, "psf_overlap");
    const size_t tab = (size_t)(nsamp + 12) * (nsamp + 12);
    for (int p0 = 0; p0 < npairs; p0 += chunk) {
        ctx->ws_used = 0;
        IMCOM_TRY(fft_inverse(ctx, pl, (const cplx *)spec1, (const cplx *)spec2, pairs_host + 2 * (size_t)p0, std::min(chunk, npairs - p0), nsamp,
                              amp_penalty, tables + (size_t)p0 * tab));
    }
    return IMCOM_OK;
}
