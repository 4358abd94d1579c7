// psf_overlap.hip -- PSF cross-correlation tables on the device.
//
// Replaces PSFGrp.accel_pad_and_rfft2 (reference src/pyimcom/psfutil.py:943-986) and
// PSFOvl._build_psfovl / accel_irfft2_and_extract (1244-1294, 1178-1242):
//     table[p,q] = roll(irfft2(rfft2(pad(psf1[p])) * conj(rfft2(pad(psf2[q])))), nc)[:nsamp, :nsamp]
// written with the 6-pixel zero border the interpolators expect.
//
// Formulation: the zero-padded 2-D DFTs are evaluated as dense real matrix products with exact
// twiddle matrices (integer argument reduction mod nfft, then cos/sin of a multiple of pi/nfft), on the
// fp64 MFMA tile engine of gemm_f64.hip.  Only the nsamp x nsamp window of the inverse transform that
// is kept is ever computed.  Cost is O(nfft * nsamp * nh) per transform stage instead of an FFT's
// O(n^2 log n); for the <= 28 tables of a 2x2 stamp group this is ~50 GFlop, amortised over the
// stamps that share the group.  (A butterfly FFT would do less work; this form is exact to fp64
// rounding and reuses the one hot GEMM kernel -- noted in DESIGN.md as the next thing to replace.)
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
